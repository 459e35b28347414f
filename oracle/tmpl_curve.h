/*
 * oracle/tmpl_curve.h -- TEST INFRASTRUCTURE ONLY.
 *
 * "Template" (included once per group) restating the short-Weierstrass a=0 Jacobian group law of
 * the reference's `curve_impl!` macro, pairing/src/bn256/ec.rs:1-631, INCLUDING its special-case
 * order.  Instantiated for G1 (base field Fq) and G2 (base field Fq2) in bn254_oracle.c.
 *
 * Required macros before inclusion:
 *   CNAME(x)            name mangler, e.g. g1_##x
 *   F                   base-field element type
 *   F_ZERO(p) F_ONE(p) F_IS_ZERO(p) F_EQ(a,b)
 *   F_ADD(r,a,b) F_SUB(r,a,b) F_DBL(r,a) F_NEG(r,a) F_MUL(r,a,b) F_SQR(r,a) F_INV(r,a)
 */

typedef struct { F x, y; } CNAME(affine_t);   /* raw layout: all-zero bytes == point at infinity (ec.rs:673-675) */
typedef struct { F x, y, z; } CNAME(jac_t);   /* Jacobian; z == 0 <=> infinity (ec.rs:227-246) */

static inline int CNAME(affine_is_zero)(const CNAME(affine_t) *p) { return F_IS_ZERO(&p->x) && F_IS_ZERO(&p->y); }
static inline int CNAME(is_zero)(const CNAME(jac_t) *p) { return F_IS_ZERO(&p->z); }

/* ec.rs:229-235: zero() = (0, 1, 0) */
static inline void CNAME(set_zero)(CNAME(jac_t) *p) { F_ZERO(&p->x); F_ONE(&p->y); F_ZERO(&p->z); }

/* ec.rs:580-592 From<affine> for projective */
static inline void CNAME(from_affine)(CNAME(jac_t) *r, const CNAME(affine_t) *p) {
  if (CNAME(affine_is_zero)(p)) { CNAME(set_zero)(r); return; }
  r->x = p->x; r->y = p->y; F_ONE(&r->z);
}

/* ec.rs:301-358 double(): dbl-2009-l */
static void CNAME(double)(CNAME(jac_t) *p) {
  if (CNAME(is_zero)(p)) return;
  F a, b, c, d, e, f;
  F_SQR(&a, &p->x);
  F_SQR(&b, &p->y);
  F_SQR(&c, &b);
  F_ADD(&d, &p->x, &b);
  F_SQR(&d, &d);
  F_SUB(&d, &d, &a);
  F_SUB(&d, &d, &c);
  F_DBL(&d, &d);
  F_DBL(&e, &a);
  F_ADD(&e, &e, &a);
  F_SQR(&f, &e);
  F_MUL(&p->z, &p->z, &p->y);
  F_DBL(&p->z, &p->z);
  F_SUB(&p->x, &f, &d);
  F_SUB(&p->x, &p->x, &d);
  F_SUB(&p->y, &d, &p->x);
  F_MUL(&p->y, &p->y, &e);
  F_DBL(&c, &c); F_DBL(&c, &c); F_DBL(&c, &c);
  F_SUB(&p->y, &p->y, &c);
}

/* ec.rs:360-454 add_assign(): add-2007-bl */
static void CNAME(add)(CNAME(jac_t) *p, const CNAME(jac_t) *o) {
  if (CNAME(is_zero)(p)) { *p = *o; return; }
  if (CNAME(is_zero)(o)) return;
  F z1z1, z2z2, u1, u2, s1, s2;
  F_SQR(&z1z1, &p->z);
  F_SQR(&z2z2, &o->z);
  F_MUL(&u1, &p->x, &z2z2);
  F_MUL(&u2, &o->x, &z1z1);
  F_MUL(&s1, &p->y, &o->z); F_MUL(&s1, &s1, &z2z2);
  F_MUL(&s2, &o->y, &p->z); F_MUL(&s2, &s2, &z1z1);
  if (F_EQ(&u1, &u2) && F_EQ(&s1, &s2)) { CNAME(double)(p); return; }
  if (F_EQ(&u1, &u2)) { CNAME(set_zero)(p); return; }
  F h, i, j, r, v;
  F_SUB(&h, &u2, &u1);
  F_DBL(&i, &h); F_SQR(&i, &i);
  F_MUL(&j, &h, &i);
  F_SUB(&r, &s2, &s1); F_DBL(&r, &r);
  F_MUL(&v, &u1, &i);
  F_SQR(&p->x, &r);
  F_SUB(&p->x, &p->x, &j);
  F_SUB(&p->x, &p->x, &v);
  F_SUB(&p->x, &p->x, &v);
  F_SUB(&p->y, &v, &p->x);
  F_MUL(&p->y, &p->y, &r);
  F_MUL(&s1, &s1, &j); F_DBL(&s1, &s1);
  F_SUB(&p->y, &p->y, &s1);
  F_ADD(&p->z, &p->z, &o->z);
  F_SQR(&p->z, &p->z);
  F_SUB(&p->z, &p->z, &z1z1);
  F_SUB(&p->z, &p->z, &z2z2);
  F_MUL(&p->z, &p->z, &h);
}

/* ec.rs:456-536 add_assign_mixed(): madd-2007-bl.  NB the reference has no explicit P + (-P)
 * branch here: H == 0 makes Z3 == 0 (ec.rs:487), i.e. the result is infinity with garbage x,y. */
static void CNAME(add_mixed)(CNAME(jac_t) *p, const CNAME(affine_t) *o) {
  if (CNAME(affine_is_zero)(o)) return;
  if (CNAME(is_zero)(p)) { p->x = o->x; p->y = o->y; F_ONE(&p->z); return; }
  F z1z1, u2, s2;
  F_SQR(&z1z1, &p->z);
  F_MUL(&u2, &o->x, &z1z1);
  F_MUL(&s2, &o->y, &p->z); F_MUL(&s2, &s2, &z1z1);
  if (F_EQ(&p->x, &u2) && F_EQ(&p->y, &s2)) { CNAME(double)(p); return; }
  F h, hh, i, j, r, v;
  F_SUB(&h, &u2, &p->x);
  F_SQR(&hh, &h);
  F_DBL(&i, &hh); F_DBL(&i, &i);
  F_MUL(&j, &h, &i);
  F_SUB(&r, &s2, &p->y); F_DBL(&r, &r);
  F_MUL(&v, &p->x, &i);
  F_SQR(&p->x, &r);
  F_SUB(&p->x, &p->x, &j);
  F_SUB(&p->x, &p->x, &v);
  F_SUB(&p->x, &p->x, &v);
  F_MUL(&j, &j, &p->y); F_DBL(&j, &j);
  F_SUB(&p->y, &v, &p->x);
  F_MUL(&p->y, &p->y, &r);
  F_SUB(&p->y, &p->y, &j);
  F_ADD(&p->z, &p->z, &h);
  F_SQR(&p->z, &p->z);
  F_SUB(&p->z, &p->z, &z1z1);
  F_SUB(&p->z, &p->z, &hh);
}

/* ec.rs:538-542 */
static inline void CNAME(negate)(CNAME(jac_t) *p) { if (!CNAME(is_zero)(p)) F_NEG(&p->y, &p->y); }

/* ec.rs:544-563 mul_assign(): MSB-first double-and-add over the 256 bits of a canonical FrRepr */
static void CNAME(mul)(CNAME(jac_t) *p, const uint64_t k[4]) {
  CNAME(jac_t) res;
  CNAME(set_zero)(&res);
  int found_one = 0;
  for (int i = 255; i >= 0; --i) {
    int bit = (int)((k[i / 64] >> (i % 64)) & 1);
    if (found_one) CNAME(double)(&res); else found_one = bit;
    if (bit) CNAME(add)(&res, p);
  }
  *p = res;
}

/* ec.rs:596-629 From<projective> for affine.  Infinity -> all-zero raw record (boundary convention,
 * the library treats only all-zero as infinity). */
static void CNAME(to_affine)(CNAME(affine_t) *r, const CNAME(jac_t) *p) {
  if (CNAME(is_zero)(p)) { F_ZERO(&r->x); F_ZERO(&r->y); return; }
  F one; F_ONE(&one);
  if (F_EQ(&p->z, &one)) { r->x = p->x; r->y = p->y; return; }
  F zinv, zp;
  F_INV(&zinv, &p->z);
  F_SQR(&zp, &zinv);
  F_MUL(&r->x, &p->x, &zp);
  F_MUL(&zp, &zp, &zinv);
  F_MUL(&r->y, &p->y, &zp);
}

/* ec.rs:45-85 PartialEq for projective (cross-multiplied comparison) */
static int CNAME(eq)(const CNAME(jac_t) *a, const CNAME(jac_t) *b) {
  if (CNAME(is_zero)(a)) return CNAME(is_zero)(b);
  if (CNAME(is_zero)(b)) return 0;
  F z1, z2, t1, t2;
  F_SQR(&z1, &a->z);
  F_SQR(&z2, &b->z);
  F_MUL(&t1, &a->x, &z2);
  F_MUL(&t2, &b->x, &z1);
  if (!F_EQ(&t1, &t2)) return 0;
  F_MUL(&z1, &z1, &a->z);
  F_MUL(&z2, &z2, &b->z);
  F_MUL(&z2, &z2, &a->y);
  F_MUL(&z1, &z1, &b->y);
  return F_EQ(&z1, &z2);
}

/* ec.rs:251-299 batch_normalization (Montgomery's trick): x/z^2, y/z^3, z = one for every
 * element that is neither infinity nor already normalized. */
static void CNAME(batch_normalization)(CNAME(jac_t) *v, size_t n) {
  F one; F_ONE(&one);
  F *prod = (F *)malloc((n ? n : 1) * sizeof(F));
  unsigned char *todo = (unsigned char *)malloc(n ? n : 1);
  F tmp = one;
  size_t cnt = 0;
  for (size_t i = 0; i < n; ++i) {                 /* first pass ec.rs:258-267: [a, ab, abc, ...] */
    todo[i] = !(CNAME(is_zero)(&v[i]) || F_EQ(&v[i].z, &one));
    if (!todo[i]) continue;
    F_MUL(&tmp, &tmp, &v[i].z);
    prod[cnt++] = tmp;
  }
  if (cnt) F_INV(&tmp, &tmp);                      /* ec.rs:270 */
  size_t k = cnt;
  for (size_t ii = n; ii-- > 0;) {                 /* second pass ec.rs:273-286 (backwards) */
    if (!todo[ii]) continue;
    --k;
    F s = (k == 0) ? one : prod[k - 1];
    F newtmp;
    F_MUL(&newtmp, &tmp, &v[ii].z);
    F_MUL(&v[ii].z, &tmp, &s);
    tmp = newtmp;
  }
  for (size_t i = 0; i < n; ++i) {                 /* third pass ec.rs:289-298 */
    if (!todo[i]) continue;
    F z;
    F_SQR(&z, &v[i].z);
    F_MUL(&v[i].x, &v[i].x, &z);
    F_MUL(&z, &z, &v[i].z);
    F_MUL(&v[i].y, &v[i].y, &z);
    v[i].z = one;
  }
  free(prod);
  free(todo);
}

/*
 * oracle/tmpl_multiexp.h -- TEST INFRASTRUCTURE ONLY.
 *
 * "Template" restating bellman/src/multiexp.rs:53-157 (`multiexp_inner`) and :330-355 (`multiexp`)
 * together with the `(Arc<Vec<G>>, usize)` Source of bellman/src/source.rs:36-70 and the
 * QueryDensity iteration of source.rs:72-118.
 *
 * Required macros before inclusion:
 *   MNAME(x)                   name mangler
 *   M_AFFINE M_PROJ            base / accumulator types
 *   M_AFFINE_IS_ZERO(p)        base is the identity (source.rs:50)
 *   M_SET_ZERO(p) M_ADD_MIXED(p,a) M_ADD(p,o) M_DOUBLE(p)
 *   M_SCALAR_LIMBS             number of u64 limbs of the exponent repr (4 for FrRepr, 1 for DummyEngine)
 *   M_NUM_BITS                 Fr::NUM_BITS (254 for BN254, 16 for the DummyEngine)
 *
 * Error codes (== include/mi355zk.h): 0 ok, 1 UnexpectedIdentity (source.rs:50-52),
 * 2 UnexpectedEof (source.rs:46-48,62-64).
 *
 * Determinism note: in the reference every window runs as its own CpuPool task and the windows are
 * `join`ed (multiexp.rs:75,145), so when different windows fail with different errors the one that
 * surfaces is scheduling dependent.  The oracle (and the library) define it: the error at the lowest
 * exponent index wins; at one index Eof is tested before identity (source.rs:45-52).
 */

typedef struct {
  const M_AFFINE *bases; size_t n_bases; size_t cursor;
} MNAME(source_t);

/* source.rs:44-59 */
static inline int MNAME(src_add_assign_mixed)(MNAME(source_t) *s, M_PROJ *to) {
  if (s->n_bases <= s->cursor) return 2;
  if (M_AFFINE_IS_ZERO(&s->bases[s->cursor])) return 1;
  M_ADD_MIXED(to, &s->bases[s->cursor]);
  s->cursor += 1;
  return 0;
}
/* source.rs:61-69 */
static inline int MNAME(src_skip)(MNAME(source_t) *s, size_t amt) {
  if (s->n_bases <= s->cursor) return 2;
  s->cursor += amt;
  return 0;
}

static inline int MNAME(density_get)(const uint32_t *density, size_t i) {
  return density == NULL ? 1 : (int)((density[i / 32] >> (i % 32)) & 1);
}

static inline int MNAME(repr_is)(const uint64_t *e, uint64_t v) {
  if (e[0] != v) return 0;
  for (int k = 1; k < M_SCALAR_LIMBS; ++k) if (e[k]) return 0;
  return 1;
}

/* (exp >> skip).as_ref()[0] % (1 << c), multiexp.rs:110-111 */
static inline uint64_t MNAME(window_digit)(const uint64_t *e, uint32_t skip, uint32_t c) {
  uint32_t limb = skip / 64, off = skip % 64;
  uint64_t lo = limb < M_SCALAR_LIMBS ? e[limb] >> off : 0;
  if (off && limb + 1 < M_SCALAR_LIMBS) lo |= e[limb + 1] << (64 - off);
  return lo % ((uint64_t)1 << c);
}

/* One region (window) of multiexp_inner, multiexp.rs:75-133.  `err_index` receives the exponent
 * index at which an error was raised. */
static int MNAME(window)(const M_AFFINE *bases, size_t n_bases, size_t base_offset,
                         const uint64_t *exps, size_t n, const uint32_t *density,
                         uint32_t skip, uint32_t c, int handle_trivial, M_PROJ *out, size_t *err_index) {
  M_PROJ acc;
  M_SET_ZERO(&acc);
  MNAME(source_t) src = {bases, n_bases, base_offset};
  size_t nb = ((size_t)1 << c) - 1;
  M_PROJ *buckets = (M_PROJ *)malloc(nb * sizeof(M_PROJ));
  for (size_t i = 0; i < nb; ++i) M_SET_ZERO(&buckets[i]);
  int rc = 0;
  for (size_t i = 0; i < n && rc == 0; ++i) {
    if (!MNAME(density_get)(density, i)) continue;
    const uint64_t *e = exps + i * M_SCALAR_LIMBS;
    if (MNAME(repr_is)(e, 0)) {
      rc = MNAME(src_skip)(&src, 1);
    } else if (MNAME(repr_is)(e, 1)) {
      rc = handle_trivial ? MNAME(src_add_assign_mixed)(&src, &acc) : MNAME(src_skip)(&src, 1);
    } else {
      uint64_t d = MNAME(window_digit)(e, skip, c);
      rc = d ? MNAME(src_add_assign_mixed)(&src, &buckets[d - 1]) : MNAME(src_skip)(&src, 1);
    }
    if (rc) *err_index = i;
  }
  if (rc == 0) {
    /* summation by parts, multiexp.rs:122-130 */
    M_PROJ running;
    M_SET_ZERO(&running);
    for (size_t k = nb; k-- > 0;) {
      M_ADD(&running, &buckets[k]);
      M_ADD(&acc, &running);
    }
    *out = acc;
  }
  free(buckets);
  return rc;
}

/* multiexp.rs:341-345 */
static inline uint32_t MNAME(choose_c)(size_t n) {
#ifdef ORACLE_WINDOW_OVERRIDE
  /* BENCH ONLY (oracle_multiexp_set_window_bits): the width the reference would pick at ANOTHER size -- the 2^26 headline's c = 19 timed
   * on a 2^22 sample; 0 = the reference's rule below */
  if (ORACLE_WINDOW_OVERRIDE) return ORACLE_WINDOW_OVERRIDE;
#endif
  if (n < 32) return 3;
  return (uint32_t)ceil(log((double)(uint32_t)n));
}

typedef struct {
  const M_AFFINE *bases; size_t n_bases, base_offset;
  const uint64_t *exps; size_t n; const uint32_t *density;
  uint32_t c, n_windows; M_PROJ *results; int *rcs; size_t *err_idx;
  volatile int next;
} MNAME(job_t);

static void *MNAME(worker)(void *arg) {
  MNAME(job_t) *job = (MNAME(job_t) *)arg;
  for (;;) {
    int w = __sync_fetch_and_add(&job->next, 1);
    if (w >= (int)job->n_windows) break;
    job->err_idx[w] = 0;
    job->rcs[w] = MNAME(window)(job->bases, job->n_bases, job->base_offset, job->exps, job->n, job->density,
                                (uint32_t)w * job->c, job->c, w == 0, &job->results[w], &job->err_idx[w]);
  }
  return NULL;
}

/* multiexp(), multiexp.rs:330-355 + the join of multiexp_inner :136-155.
 * n = min(n_scalars, density_bits) (the `zip` at multiexp.rs:92).  `threads` mirrors the Worker's
 * CpuPool: one task per window (at most ceil(NUM_BITS/c) run concurrently). */
static int MNAME(multiexp)(const M_AFFINE *bases, size_t n_bases, size_t base_offset,
                           const uint64_t *exps, size_t n_scalars,
                           const uint32_t *density, size_t density_bits, int threads, M_PROJ *out) {
  size_t n = n_scalars;
  if (density != NULL && density_bits < n) n = density_bits;
  uint32_t c = MNAME(choose_c)(n_scalars);
  uint32_t n_windows = (M_NUM_BITS + c - 1) / c; /* skip = 0, c, 2c, ... while skip < NUM_BITS */
  M_PROJ *results = (M_PROJ *)malloc(n_windows * sizeof(M_PROJ));
  int *rcs = (int *)calloc(n_windows, sizeof(int));
  size_t *err_idx = (size_t *)calloc(n_windows, sizeof(size_t));
  MNAME(job_t) job = {bases, n_bases, base_offset, exps, n, density, c, n_windows, results, rcs, err_idx, 0};
  if (threads <= 1) {
    MNAME(worker)(&job);
  } else {
    if (threads > (int)n_windows) threads = (int)n_windows;
    pthread_t *tid = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    for (int t = 0; t < threads; ++t) pthread_create(&tid[t], NULL, MNAME(worker), &job);
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    free(tid);
  }
  int rc = 0;
  size_t best = (size_t)-1;
  for (uint32_t w = 0; w < n_windows; ++w)
    if (rcs[w] && err_idx[w] < best) { best = err_idx[w]; rc = rcs[w]; }
  if (rc == 0) {
    /* higher.double() x c; higher += this  (multiexp.rs:146-154), from the top window down */
    M_PROJ acc = results[n_windows - 1];
    for (uint32_t w = n_windows - 1; w-- > 0;) {
      for (uint32_t k = 0; k < c; ++k) M_DOUBLE(&acc);
      M_ADD(&acc, &results[w]);
    }
    *out = acc;
  }
  free(results); free(rcs); free(err_idx);
  return rc;
}

/* ---- powersoftau's dense_multiexp (powersoftau/src/utils.rs:189-292): bases.len() == exponents.len(), no Source, no density.
 * ALL cores work on ONE region at a time: the bases are cut into chunks of n / cpus + 1 (:217), every chunk gets a thread with its
 * own bucket array (:227) that fills it (zero exponents skipped, exponent one added directly in region 0, :243-258), sums it by
 * parts (:262-266) and adds its result to the region's sum under a lock (:268-270); the regions are joined as in multiexp_inner
 * (:276-291: the higher region doubled c times, plus this one).  A base at infinity adds nothing (add_assign_mixed returns at
 * once for an identity operand, ec.rs:457-459): there is no error path. */
typedef struct {
  const M_AFFINE *bases; const uint64_t *exps; size_t lo, hi;
  uint32_t skip, c; int handle_trivial; M_PROJ acc;
} MNAME(dense_job_t);

static void *MNAME(dense_worker)(void *arg) {
  MNAME(dense_job_t) *j = (MNAME(dense_job_t) *)arg;
  size_t nb = ((size_t)1 << j->c) - 1;
  M_PROJ *buckets = (M_PROJ *)malloc(nb * sizeof(M_PROJ));
  for (size_t i = 0; i < nb; ++i) M_SET_ZERO(&buckets[i]);
  M_PROJ acc;
  M_SET_ZERO(&acc);
  for (size_t i = j->lo; i < j->hi; ++i) {
    const uint64_t *e = j->exps + i * M_SCALAR_LIMBS;
    if (MNAME(repr_is)(e, 0)) continue;
    if (MNAME(repr_is)(e, 1)) {
      if (j->handle_trivial && !M_AFFINE_IS_ZERO(&j->bases[i])) M_ADD_MIXED(&acc, &j->bases[i]);
    } else {
      uint64_t d = MNAME(window_digit)(e, j->skip, j->c);
      if (d && !M_AFFINE_IS_ZERO(&j->bases[i])) M_ADD_MIXED(&buckets[d - 1], &j->bases[i]);
    }
  }
  M_PROJ running;
  M_SET_ZERO(&running);
  for (size_t k = nb; k-- > 0;) {
    M_ADD(&running, &buckets[k]);
    M_ADD(&acc, &running);
  }
  j->acc = acc;
  free(buckets);
  return NULL;
}

static void MNAME(dense_multiexp)(const M_AFFINE *bases, const uint64_t *exps, size_t n, int cpus, M_PROJ *out) {
  uint32_t c = MNAME(choose_c)(n);  /* utils.rs:199-203: the same rule as multiexp */
  if (cpus < 1) cpus = 1;
  size_t chunk = n / (size_t)cpus + 1;
  size_t n_jobs = (n + chunk - 1) / chunk;
  if (n_jobs == 0) n_jobs = 1;
  uint32_t n_regions = (M_NUM_BITS + c - 1) / c;
  M_PROJ *regions = (M_PROJ *)malloc(n_regions * sizeof(M_PROJ));
  MNAME(dense_job_t) *jobs = (MNAME(dense_job_t) *)malloc(n_jobs * sizeof(MNAME(dense_job_t)));
  pthread_t *tid = (pthread_t *)malloc(n_jobs * sizeof(pthread_t));
  for (uint32_t r = 0; r < n_regions; ++r) {
    for (size_t t = 0; t < n_jobs; ++t) {
      size_t lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
      if (lo > n) lo = n;
      jobs[t].bases = bases; jobs[t].exps = exps; jobs[t].lo = lo; jobs[t].hi = hi;
      jobs[t].skip = r * c; jobs[t].c = c; jobs[t].handle_trivial = r == 0;
      if (n_jobs > 1) pthread_create(&tid[t], NULL, MNAME(dense_worker), &jobs[t]);
      else MNAME(dense_worker)(&jobs[t]);
    }
    M_SET_ZERO(&regions[r]);
    for (size_t t = 0; t < n_jobs; ++t) {
      if (n_jobs > 1) pthread_join(tid[t], NULL);
      M_ADD(&regions[r], &jobs[t].acc);  /* (the reference adds in completion order under a Mutex: the same group element) */
    }
  }
  M_PROJ acc = regions[n_regions - 1];
  for (uint32_t r = n_regions - 1; r-- > 0;) {
    for (uint32_t k = 0; k < c; ++k) M_DOUBLE(&acc);
    M_ADD(&acc, &regions[r]);
  }
  *out = acc;
  free(regions); free(jobs); free(tid);
}

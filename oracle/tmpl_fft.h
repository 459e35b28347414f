/*
 * oracle/tmpl_fft.h -- TEST INFRASTRUCTURE ONLY.
 *
 * "Template" restating bellman/src/domain.rs: `serial_fft` :274-317 (the normative definition),
 * `parallel_fft` :319-376, `best_fft` :263-272, and the EvaluationDomain wrappers `fft` :154,
 * `ifft` :159-173, `distribute_powers` :176-189, `coset_fft` :191-195, `icoset_fft` :197-203,
 * plus the constants `from_coeffs` derives (:52-99).  Elements are `Scalar<E>` (group.rs:53-82),
 * i.e. plain field elements.
 *
 * Required macros before inclusion:
 *   TNAME(x)        name mangler
 *   T_FE            element type (`G: Group<E>`, bellman/src/group.rs:15-20): Scalar<E> (group.rs:53-82) or
 *                   Point<G> (group.rs:22-51, the curve-point FFT of powersoftau/src/bin/prepare_phase2.rs:68-105)
 *   T_ZERO(p) T_ADD(r,a,b) T_SUB(r,a,b)      group_zero / group_add_assign / group_sub_assign
 *   T_MULS(r,a,s)                            group_mul_assign(&s): element times SCALAR
 *   S_FE            scalar field element type (E::Fr)
 *   S_ONE(p) S_MUL(r,a,b) S_INV(r,a) S_POW64(r,a,e)   scalar arithmetic (twiddles, domain constants)
 *   S_ROOT_OF_UNITY(p) S_GENERATOR(p) S_FROM_U64(p,v) S_S   Fr::root_of_unity / multiplicative_generator / from_str / S
 */

static uint32_t TNAME(bitreverse)(uint32_t n, uint32_t l) { /* domain.rs:276-283 */
  uint32_t r = 0;
  for (uint32_t i = 0; i < l; ++i) { r = (r << 1) | (n & 1); n >>= 1; }
  return r;
}

/* domain.rs:274-317 */
static void TNAME(serial_fft)(T_FE *a, const S_FE *omega, uint32_t log_n) {
  uint32_t n = (uint32_t)1 << log_n;
  for (uint32_t k = 0; k < n; ++k) {
    uint32_t rk = TNAME(bitreverse)(k, log_n);
    if (k < rk) { T_FE t = a[rk]; a[rk] = a[k]; a[k] = t; }
  }
  uint32_t m = 1;
  for (uint32_t s = 0; s < log_n; ++s) {
    S_FE w_m;
    S_POW64(&w_m, omega, (uint64_t)(n / (2 * m)));
    for (uint32_t k = 0; k < n; k += 2 * m) {
      S_FE w;
      S_ONE(&w);
      for (uint32_t j = 0; j < m; ++j) {
        T_FE t, tmp;
        T_MULS(&t, &a[k + j + m], &w);
        T_SUB(&tmp, &a[k + j], &t);
        a[k + j + m] = tmp;
        T_ADD(&a[k + j], &a[k + j], &t);
        S_MUL(&w, &w, &w_m);
      }
    }
    m *= 2;
  }
}

typedef struct {
  const T_FE *a; T_FE *tmp; const S_FE *omega; const S_FE *new_omega;
  uint32_t log_n, log_cpus, j;
} TNAME(pfft_job_t);

/* body of one spawned thread of parallel_fft, domain.rs:337-357 */
static void *TNAME(pfft_thread)(void *arg) {
  TNAME(pfft_job_t) *job = (TNAME(pfft_job_t) *)arg;
  uint32_t log_new_n = job->log_n - job->log_cpus;
  uint32_t num_cpus = (uint32_t)1 << job->log_cpus;
  S_FE omega_j, omega_step, elt;
  S_POW64(&omega_j, job->omega, (uint64_t)job->j);
  S_POW64(&omega_step, job->omega, (uint64_t)job->j << log_new_n);
  S_ONE(&elt);
  for (uint32_t i = 0; i < ((uint32_t)1 << log_new_n); ++i) {
    T_ZERO(&job->tmp[i]);
    for (uint32_t s = 0; s < num_cpus; ++s) {
      uint32_t idx = (i + (s << log_new_n)) % ((uint32_t)1 << job->log_n);
      T_FE t;
      T_MULS(&t, &job->a[idx], &elt);
      T_ADD(&job->tmp[i], &job->tmp[i], &t);
      S_MUL(&elt, &elt, &omega_step);
    }
    S_MUL(&elt, &elt, &omega_j);
  }
  TNAME(serial_fft)(job->tmp, job->new_omega, log_new_n);
  return NULL;
}

/* domain.rs:319-376; one pthread per sub-FFT like the reference's scope.spawn */
static void TNAME(parallel_fft)(T_FE *a, const S_FE *omega, uint32_t log_n, uint32_t log_cpus) {
  uint32_t num_cpus = (uint32_t)1 << log_cpus;
  uint32_t log_new_n = log_n - log_cpus;
  size_t sub = (size_t)1 << log_new_n;
  T_FE *tmp = (T_FE *)malloc(sizeof(T_FE) * sub * num_cpus);
  S_FE new_omega;
  S_POW64(&new_omega, omega, (uint64_t)num_cpus);
  TNAME(pfft_job_t) *jobs = (TNAME(pfft_job_t) *)malloc(sizeof(TNAME(pfft_job_t)) * num_cpus);
  pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * num_cpus);
  for (uint32_t j = 0; j < num_cpus; ++j) {
    TNAME(pfft_job_t) jb = {a, tmp + (size_t)j * sub, omega, &new_omega, log_n, log_cpus, j};
    jobs[j] = jb;
    pthread_create(&tid[j], NULL, TNAME(pfft_thread), &jobs[j]);
  }
  for (uint32_t j = 0; j < num_cpus; ++j) pthread_join(tid[j], NULL);
  uint32_t mask = num_cpus - 1;
  for (size_t idx = 0; idx < ((size_t)1 << log_n); ++idx) /* domain.rs:362-375 */
    a[idx] = tmp[(size_t)(idx & mask) * sub + (idx >> log_cpus)];
  free(tmp); free(jobs); free(tid);
}

/* domain.rs:263-272 */
static void TNAME(best_fft)(T_FE *a, const S_FE *omega, uint32_t log_n, uint32_t log_cpus) {
  if (log_n <= log_cpus) TNAME(serial_fft)(a, omega, log_n);
  else TNAME(parallel_fft)(a, omega, log_n, log_cpus);
}

typedef struct { uint32_t exp; S_FE omega, omegainv, geninv, minv; } TNAME(domain_t);

/* constants of from_coeffs for a domain of size m = 2^exp, domain.rs:61-98.
 * returns 0, or -1 for PolynomialDegreeTooLarge (exp > S, :75-77). */
static int TNAME(domain_init)(TNAME(domain_t) *d, uint32_t exp) {
  if (exp > S_S) return -1;
  d->exp = exp;
  S_ROOT_OF_UNITY(&d->omega);
  for (uint32_t i = exp; i < S_S; ++i) S_MUL(&d->omega, &d->omega, &d->omega);
  S_INV(&d->omegainv, &d->omega);
  S_FE g; S_GENERATOR(&g);
  S_INV(&d->geninv, &g);
  S_FE m; S_FROM_U64(&m, (uint64_t)1 << exp);
  S_INV(&d->minv, &m);
  return 0;
}

/* domain.rs:176-189.  The reference seeds each chunk with g^(i*chunk) and then runs a product;
 * every element ends up multiplied by exactly g^index, which is what is restated here. */
static void TNAME(distribute_powers)(T_FE *a, size_t n, const S_FE *g) {
  S_FE u; S_ONE(&u);
  for (size_t i = 0; i < n; ++i) { T_MULS(&a[i], &a[i], &u); S_MUL(&u, &u, g); }
}

static void TNAME(fft)(T_FE *a, const TNAME(domain_t) *d, uint32_t log_cpus) { TNAME(best_fft)(a, &d->omega, d->exp, log_cpus); }
static void TNAME(ifft)(T_FE *a, const TNAME(domain_t) *d, uint32_t log_cpus) {
  TNAME(best_fft)(a, &d->omegainv, d->exp, log_cpus);
  for (size_t i = 0; i < ((size_t)1 << d->exp); ++i) T_MULS(&a[i], &a[i], &d->minv);
}
static void TNAME(coset_fft)(T_FE *a, const TNAME(domain_t) *d, uint32_t log_cpus) {
  S_FE g; S_GENERATOR(&g);
  TNAME(distribute_powers)(a, (size_t)1 << d->exp, &g);
  TNAME(fft)(a, d, log_cpus);
}
static void TNAME(icoset_fft)(T_FE *a, const TNAME(domain_t) *d, uint32_t log_cpus) {
  TNAME(ifft)(a, d, log_cpus);
  TNAME(distribute_powers)(a, (size_t)1 << d->exp, &d->geninv);
}

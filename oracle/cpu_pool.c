/* cpu_pool.c - the reference's stage-parallel single-prime product with the thread handling tuned.
 *
 * TEST / BASELINE INFRASTRUCTURE ONLY (like the rest of oracle/): used by tests/ and by the cpu_baseline leg
 * of bench.py as a second, fairer CPU figure.  The product package never loads it.
 *
 * What it restates: the single-prime Montgomery programs.  monttry2.cpp:283-316 (ntt_iter_parallel) creates and
 * joins T pthreads for EVERY butterfly stage and deals the blocks of a stage to them round-robin
 * (ntt_worker, monttry2.cpp:259-280); pthread_normal.cpp:285-349 (ntt_iter_pthread) does the same with 8 fixed
 * threads on contiguous ranges and falls back to one thread once a stage has fewer than 8 blocks
 * (pthread_normal.cpp:311-325).  Arithmetic as there: Montgomery form with R = 2^30 (monttry2.cpp:376), REDC
 * (monttry2.cpp:91-100), get_result's order of operations (monttry2.cpp:319-343), conversions to and from
 * Montgomery form around it (monttry2.cpp:378-382, 390-393).
 *
 * What is tuned, and nothing else: ONE pool of threads lives for the whole product and meets at a barrier between
 * stages instead of being created and joined 18 times per transform; stages with fewer blocks than threads split
 * each block's butterflies over the threads (every thread starts its twiddle at w_len^(first j) by one modular
 * power) instead of leaving most threads idle or falling back to one; the stage roots are computed once per stage
 * by thread 0 rather than by every thread.  The butterflies themselves are the reference's.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

typedef struct {
  uint64_t N, R, Ninv_neg, R2;
  int logR;
} mont_t;

/* MontMul's constructor, monttry2.cpp:71-88: -N^-1 mod R and R^2 mod N (R = 2^30, N odd prime below R) */
static int mont_init(mont_t *m, uint64_t R, uint64_t N) {
  if (R == 0 || (R & (R - 1)) || N >= R || !(N & 1)) return -1;
  m->N = N; m->R = R; m->logR = 0;
  while ((1ULL << m->logR) < R) m->logR++;
  uint64_t inv = 1;                                     /* Newton iteration for N^-1 mod 2^k (the reference uses egcd) */
  for (int i = 0; i < 6; ++i) inv *= 2 - N * inv;
  inv &= R - 1;
  m->Ninv_neg = (R - inv) & (R - 1);
  m->R2 = (uint64_t)(((u128)(R % N) * (R % N)) % N);
  return 0;
}
/* REDC, monttry2.cpp:91-100 */
static inline uint64_t redc(const mont_t *m, uint64_t T) {
  const uint64_t mm = ((T & (m->R - 1)) * m->Ninv_neg) & (m->R - 1);
  uint64_t t = (T + mm * m->N) >> m->logR;
  if (t >= m->N) t -= m->N;
  return t;
}
static inline uint64_t to_mont(const mont_t *m, uint64_t a) { return redc(m, a * m->R2); }       /* monttry2.cpp:103-106 */
static inline uint64_t mul_mont(const mont_t *m, uint64_t a, uint64_t b) { return redc(m, a * b); } /* monttry2.cpp:115-118 */

static uint64_t pow_mod(uint64_t a, uint64_t e, uint64_t p) {                                     /* quick_mod, monttry2.cpp:198-212 */
  uint64_t r = 1 % p;
  a %= p;
  while (e) {
    if (e & 1) r = (uint64_t)((u128)r * a % p);
    a = (uint64_t)((u128)a * a % p);
    e >>= 1;
  }
  return r;
}

typedef struct {
  pthread_barrier_t bar;
  int nthreads;
  mont_t mont;
  int len, p;
  long long *va, *vb, *vc;        /* Montgomery-form vectors (long long as in the reference's vector<long long>) */
  const int *a, *b;
  int *ab;
  int n;
  uint64_t wlen_plain;            /* the current stage's root w_len, ordinary form (set by thread 0) */
} job_t;

typedef struct { job_t *job; int tid; } arg_t;

/* bit-reversal permutation, monttry2.cpp:288-293, with the index range dealt to the threads */
static void bitrev_part(long long *v, int len, int tid, int nt) {
  int bits = 0;
  while ((1 << bits) < len) ++bits;
  const int lo = (int)((long long)len * tid / nt), hi = (int)((long long)len * (tid + 1) / nt);
  for (int i = lo; i < hi; ++i) {
    int r = 0;
    for (int k = 0; k < bits; ++k) r |= ((i >> k) & 1) << (bits - 1 - k);
    if (i < r) { long long t = v[i]; v[i] = v[r]; v[r] = t; }
  }
}

/* one transform: the stages of ntt_iter_parallel (monttry2.cpp:296-315) on a standing pool */
static void transform(job_t *J, long long *v, int invert, int tid) {
  const int nt = J->nthreads, n = J->len, p = J->p;
  const mont_t *M = &J->mont;
  bitrev_part(v, n, tid, nt);
  pthread_barrier_wait(&J->bar);
  for (int len = 2; len <= n; len <<= 1) {
    if (tid == 0) {                                    /* monttry2.cpp:261-264: wn = 3^((p-1)/len), inverted for the inverse */
      uint64_t wn = pow_mod(3, (uint64_t)(p - 1) / (uint64_t)len, (uint64_t)p);
      if (invert) wn = pow_mod(wn, (uint64_t)p - 2, (uint64_t)p);
      J->wlen_plain = wn;
    }
    pthread_barrier_wait(&J->bar);
    const uint64_t wn_plain = J->wlen_plain;
    const uint64_t wnR = to_mont(M, wn_plain);
    const int half = len >> 1, blocks = n / len;
    if (blocks >= nt) {
      /* enough blocks: thread tid takes blocks tid, tid + nt, ... (monttry2.cpp:268) */
      for (int blk = tid; blk < blocks; blk += nt) {
        long long *x = v + (size_t)blk * len;
        uint64_t w = to_mont(M, 1);
        for (int j = 0; j < half; ++j) {               /* monttry2.cpp:270-277 */
          const uint64_t u = (uint64_t)x[j], t = mul_mont(M, (uint64_t)x[j + half], w);
          x[j] = (long long)((u + t) % (uint64_t)p);
          x[j + half] = (long long)((u + (uint64_t)p - t) % (uint64_t)p);
          w = mul_mont(M, w, wnR);
        }
      }
    } else {
      /* fewer blocks than threads: every block's j range is cut into nt pieces */
      const int lo = (int)((long long)half * tid / nt), hi = (int)((long long)half * (tid + 1) / nt);
      const uint64_t w0 = to_mont(M, pow_mod(wn_plain, (uint64_t)lo, (uint64_t)p));
      for (int blk = 0; blk < blocks; ++blk) {
        long long *x = v + (size_t)blk * len;
        uint64_t w = w0;
        for (int j = lo; j < hi; ++j) {
          const uint64_t u = (uint64_t)x[j], t = mul_mont(M, (uint64_t)x[j + half], w);
          x[j] = (long long)((u + t) % (uint64_t)p);
          x[j + half] = (long long)((u + (uint64_t)p - t) % (uint64_t)p);
          w = mul_mont(M, w, wnR);
        }
      }
    }
    pthread_barrier_wait(&J->bar);
  }
}

static void *worker(void *vp) {
  arg_t *A = (arg_t *)vp;
  job_t *J = A->job;
  const int tid = A->tid, nt = J->nthreads, len = J->len;
  const mont_t *M = &J->mont;
  const int lo = (int)((long long)len * tid / nt), hi = (int)((long long)len * (tid + 1) / nt);
  for (int i = lo; i < hi; ++i) {                      /* pad and convert, monttry2.cpp:369-382 */
    J->va[i] = (long long)to_mont(M, (uint64_t)(i < J->n ? J->a[i] : 0));
    J->vb[i] = (long long)to_mont(M, (uint64_t)(i < J->n ? J->b[i] : 0));
  }
  pthread_barrier_wait(&J->bar);
  transform(J, J->va, 0, tid);                          /* get_result, monttry2.cpp:326-327 */
  transform(J, J->vb, 0, tid);
  for (int i = lo; i < hi; ++i) J->vc[i] = (long long)mul_mont(M, (uint64_t)J->va[i], (uint64_t)J->vb[i]);   /* :329-331 */
  pthread_barrier_wait(&J->bar);
  transform(J, J->vc, 1, tid);                          /* :335 */
  const uint64_t invR = to_mont(M, pow_mod((uint64_t)len, (uint64_t)J->p - 2, (uint64_t)J->p));          /* :337-340 */
  const int nout = 2 * J->n - 1;
  for (int i = lo; i < hi; ++i) {
    const uint64_t c = mul_mont(M, (uint64_t)J->vc[i], invR);
    if (i < nout) J->ab[i] = (int)redc(M, c);           /* fromMont, monttry2.cpp:109-112, 390-393 */
  }
  return NULL;
}

/* ab[0 .. 2n-2] = a * b mod p on a pool of `threads` threads.  0, or a negative code (bad modulus, no memory). */
int pool_poly_multiply_mont(const int *a, const int *b, int *ab, int n, int p, int threads) {
  if (n < 1 || p < 3 || threads < 1 || threads > 256) return -3;
  job_t J;
  memset(&J, 0, sizeof J);
  if (mont_init(&J.mont, 1ULL << 30, (uint64_t)p)) return -1;
  int len = 1;
  while (len < 2 * n) len <<= 1;
  if (((uint64_t)p - 1) % (uint64_t)len) return -2;     /* no root of that order */
  J.len = len; J.p = p; J.n = n; J.a = a; J.b = b; J.ab = ab; J.nthreads = threads;
  J.va = (long long *)malloc((size_t)len * 3 * sizeof(long long));
  if (!J.va) return -3;
  J.vb = J.va + len; J.vc = J.vb + len;
  pthread_barrier_init(&J.bar, NULL, (unsigned)threads);
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  arg_t *args = (arg_t *)malloc(sizeof(arg_t) * (size_t)threads);
  for (int t = 1; t < threads; ++t) { args[t].job = &J; args[t].tid = t; pthread_create(&th[t], NULL, worker, &args[t]); }
  args[0].job = &J; args[0].tid = 0;
  worker(&args[0]);
  for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
  pthread_barrier_destroy(&J.bar);
  free(th); free(args); free(J.va);
  return 0;
}

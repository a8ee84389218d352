/* nttx — C ABI of the MI355X NTT polynomial-multiplication engine (libnttx.so).
 *
 * The reference has no library boundary: its hot path is the timed block between
 * fRead and fCheck in each program's main() (main.cc:283-341, main_pthread.cc:386-475,
 * monttry2.cpp:385-393, pthread_normal.cpp:450-461) plus the harness-declared entry
 * point  void poly_multiply(int *a, int *b, int *ab, int n, int p)
 * (monttry2.cpp:344-350, pthread_normal.cpp:352-358).  Each function below names the
 * reference code it stands in for.
 *
 * Conventions: every function returns 0 on success or a negative NTTX_E* code and
 * never throws; nttx_last_error() returns a message for the calling thread's last
 * failure.  Calls are synchronous unless a stream argument says otherwise.  Host
 * buffers are caller-owned; nothing is retained after a call returns.  All calls
 * of one process are serialised on an internal mutex.
 *
 * Streams: every call of the process shares one device workspace.  A call enqueued on a
 * different stream than the previous call's (host-buffer calls count as the null stream)
 * first waits on the device for the work the previous stream holds, so calls on several
 * streams are safe but do not overlap each other.  A stream handed to the library must
 * stay alive until a later call has been made on another stream, or until nttx_shutdown.
 */
#ifndef NTTX_H
#define NTTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTTX_OK 0
#define NTTX_EINVAL (-1)   /* bad argument (n <= 0, null pointer, p < 2, ...) */
#define NTTX_ERANGE (-2)   /* exact result not representable: n*(p-1)^2 >= product of CRT primes, or length unsupported */
#define NTTX_EDEVICE (-3)  /* HIP runtime failure */
#define NTTX_ENOINIT (-4)  /* nttx_init has not been called */
#define NTTX_ENOTREADY (-5) /* a call that may not allocate found no tables for its modulus and length */

/* Which arithmetic path a (p, n) pair takes. */
#define NTTX_PATH_DIRECT32 0 /* one NTT modulo p itself: p < 2^31 prime, 2^k | p-1 (monttry2.cpp get_result) */
#define NTTX_PATH_CRT4 1     /* the four 32-bit NTT primes + CRT of main.cc:245-338 (as few of them as keep
                              * n*(p-1)^2 below their product: same coefficients; coefficients at or above p
                              * are first reduced modulo p, so the bound holds for any input word) */
#define NTTX_PATH_DIRECT64 2 /* one NTT modulo p itself on 64-bit lanes: 2^31 <= p < 2^62 prime, 2^k | p-1 */

/* Select the GPU of this process and build its context.  No counterpart in the reference (CPU-only program start-up;
 * main.cc:230-270 builds the Barrett/CRT constants once per process, and so does this).  The usual arrangement is one
 * process per GPU (each calls nttx_init with its own device and exchanges data through the building blocks below). */
int nttx_init(int device);
/* One process driving several GPUs: a context (tables, workspace, streams) per listed device; devices[0] becomes the
 * current one.  What the reference spreads over host threads (one per CRT modulus, main_pthread.cc:413-428, and one per
 * index chunk, main_pthread.cc:435-465) a host program can then spread over GPUs: nttx_poly_multiply_batch_u32 deals its
 * products to all listed devices by itself (one host thread per device), and every other entry point acts on the
 * current device, which nttx_set_device switches (calls stay serialised on the host; the devices' queues run side by
 * side).  Calling it again with the same list is a no-op; another list needs nttx_shutdown first. */
int nttx_init_devices(const int *devices, int ndev);
/* Number of devices this process initialised (0 before nttx_init). */
int nttx_device_count(void);
/* Make `device` (one of the initialised ones) the device later calls act on. */
int nttx_set_device(int device);
/* The contiguous share [*lo, *hi) of `batch` independent products that device k of ndev gets from
 * nttx_poly_multiply_batch_u32 (shares differ by at most one product).  Needs no device. */
int nttx_deal_batch(int batch, int ndev, int k, int *lo, int *hi);
void nttx_shutdown(void);
const char *nttx_last_error(void);

/* Page-locked host memory for long-lived buffers (the harness's arrays a, b, ab: globals in main.cc:228 /
 * monttry2.cpp:352).  Optional: the host-buffer entry points below accept any host memory, but move page-locked
 * buffers by DMA at PCIe speed instead of through the runtime's staging of pageable memory.  Returns NULL on failure
 * (nttx_last_error() says why); needs nttx_init; release with nttx_host_free before nttx_shutdown. */
void *nttx_host_alloc(size_t bytes);
void nttx_host_free(void *ptr);

/* Path that nttx_poly_multiply_u64 / _i32 will take for this modulus and length. */
int nttx_path_for(uint64_t p, int n);

/* ab[0 .. 2n-2] = a * b mod p, canonical.  Harness entry point with (T,P) = (int,int):
 * poly_multiply monttry2.cpp:344-350 as used by main() monttry2.cpp:375-393
 * (toMont, get_result, fromMont) and pthread_normal.cpp:440-461. */
int nttx_poly_multiply_i32(const int32_t *a, const int32_t *b, int32_t *ab, int n, int32_t p);

/* Same for (T,P) = (uint64_t,int64_t): the timed block of main.cc:283-341 /
 * main_pthread.cc:386-475 (residue split, 4 x ntt_compute_omp, CRT, mod p).
 * Takes the direct single-prime path when p allows it, the 4-prime CRT path otherwise. */
int nttx_poly_multiply_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p);

/* One transform modulo p itself on 64-bit lanes (Montgomery radix 2^64) for NTT-friendly moduli
 * beyond 32 bits, e.g. 19*2^46+1 of harness id 4: the single-prime structure of get_result
 * (monttry2.cpp:319-343), which the reference cannot apply there (its int p overflows) and
 * replaces by main.cc's four primes + CRT.  Coefficients are identical to the CRT route. */
int nttx_poly_multiply_direct_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p);

/* Force the 4-prime CRT path (main.cc:286-338) whatever p is. */
int nttx_poly_multiply_crt_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p);

/* batch independent products: a, b are [batch][n], ab is [batch][2n-1]; p < 2^31 on the
 * direct path.  No counterpart in the reference (one product per call). */
int nttx_poly_multiply_batch_u32(const uint32_t *a, const uint32_t *b, uint32_t *ab, int n, uint32_t p, int batch);

/* In-place length-len transform, natural order in and out, no 1/len scaling:
 * ntt_iter_barrett(a, p, root = 3, invert) main.cc:143-182.  len a power of two dividing p-1. */
int nttx_ntt_u32(uint32_t *x, int len, uint32_t p, int inverse);

/* One single-prime cyclic convolution of zero-padded residue vectors, host buffers of len
 * words each: ntt_compute_omp main.cc:185-209 / ntt_thread_func main_pthread.cc:266-290. */
int nttx_cyclic_multiply_u32(const uint32_t *a, const uint32_t *b, uint32_t *c, int len, uint32_t p);

/* CRT recombination alone, host buffers: res[k] are the four residue vectors (len words,
 * moduli 469762049, 998244353, 1004535809, 1224736769), ab[i] = x_i mod p for i < nout.
 * main.cc:313-338 / crt_thread_func main_pthread.cc:293-310. */
int nttx_crt_combine_u64(const uint32_t *const res[4], uint64_t *ab, int len, int nout, uint64_t p);

/* ---- device-resident variants (benchmarks, pipelines that keep data in HBM) ---------- */

/* d_c[b] = cyclic product of d_a[b] and d_b[b]; all three are [batch][len] u32 in device
 * memory, inputs canonical (< p) and zero-padded by the caller, outputs canonical.
 * Inputs are left untouched.  Asynchronous on `stream` (a hipStream_t, NULL = default). */
int nttx_dev_cyclic_multiply_u32(const void *d_a, const void *d_b, void *d_c, int len, uint32_t p, int batch,
                                 void *stream);

/* d_c[b] = d_a[b] * d_b[b] mod (x^len + 1, p): the negacyclic product (wrap-around with a sign
 * change), [batch][len] u32 arrays as above; needs 2*len | p-1.  No counterpart in the reference,
 * whose products are always zero-padded (main.cc:278-281); for callers that work in
 * Z_p[x]/(x^len + 1) and would otherwise pay for a transform of twice the length. */
int nttx_dev_negacyclic_multiply_u32(const void *d_a, const void *d_b, void *d_c, int len, uint32_t p, int batch,
                                     void *stream);

/* Linear product of n-coefficient polynomials held zero-padded in [batch][len] arrays, len the
 * smallest power of two >= 2n (the padding of main.cc:278-281): d_c[b][0..2n-2] = a*b mod p, the
 * rest of d_c[b] is zero.  Words of d_a, d_b at index >= n must be zero and are not read on lengths
 * that take more than one pass (half the input traffic of the cyclic call).  Asynchronous on `stream`. */
int nttx_dev_poly_multiply_u32(const void *d_a, const void *d_b, void *d_c, int n, uint32_t p, int batch,
                               void *stream);

/* nttx_dev_poly_multiply_u32 for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference, which allocates its vectors
 * inside every product (main.cc:286-298).  Same arguments, results and stream semantics, but the two transformed
 * copies go to a fixed scratch array inside the library (2 x 16 MiB of device memory per GPU, held until the process
 * ends) instead of hipMalloc'ed workspace, and the call makes no hipMalloc / hipFree and creates no event:
 *  - batch * len (len = smallest power of two >= 2n) may not exceed 2^22 words: NTTX_ERANGE beyond that;
 *  - the tables of (p, len) must exist already, i.e. nttx_dev_poly_multiply_u32 has run once with this modulus and
 *    length at set-up: NTTX_ENOTREADY otherwise (building tables allocates);
 *  - every call on a GPU shares that GPU's scratch array and the library does not order such calls against each
 *    other: two calls in flight at once on one GPU (different streams) race; calls on one stream are safe;
 *  - per-kernel event timing (nttx_profile_enable) does not cover this entry point. */
int nttx_dev_poly_multiply_u32_static(const void *d_a, const void *d_b, void *d_c, int n, uint32_t p, int batch,
                                      void *stream);

/* d_ab[b][0..2n-2] = a*b mod p through the 4-prime CRT path; d_a, d_b are [batch][n] u64,
 * d_ab is [batch][2n-1] u64, all device memory.  Asynchronous on `stream`. */
int nttx_dev_crt_multiply_u64(const void *d_a, const void *d_b, void *d_ab, int n, uint64_t p, int batch,
                              void *stream);

/* Negacyclic product on 64-bit lanes: d_c[b] = d_a[b] * d_b[b] mod (x^len + 1, p) on [batch][len] u64 device arrays
 * for an NTT-friendly prime 2^31 <= p < 2^62 with 2*len | p-1 (the ring of lattice-based schemes; no counterpart in
 * the reference, whose products are linear).  One- and two-pass lengths apply the psi^i factors inside their first
 * and last pass, longer ones in element-wise passes around the cyclic product.  Inputs: any 64-bit words; outputs
 * canonical. */
int nttx_dev_negacyclic_multiply_u64(const void *d_a, const void *d_b, void *d_c, int len, uint64_t p, int batch, void *stream);

/* 64-bit-lane counterpart of nttx_dev_poly_multiply_u32: d_a, d_b, d_c are zero-padded
 * [batch][len] u64 arrays, len the smallest power of two >= 2n (NOT [batch][n]: every array holds batch*len words);
 * inputs below p. */
int nttx_dev_direct_multiply_u64(const void *d_a, const void *d_b, void *d_c, int n, uint64_t p, int batch,
                                 void *stream);

/* In-place batched 64-bit-lane transform, [batch][len] u64, natural order in, bit-reversed order
 * out, root 3^((p-1)/len) (or its inverse), no scaling (ntt_iter_barrett main.cc:143-182 without
 * its final permutation). */
int nttx_dev_ntt_u64(void *d_x, int len, uint64_t p, int inverse, int batch, void *stream);

/* In-place batched transform on device memory, [batch][len]; bit_reversed_out != 0 skips
 * the final permutation (forward output stays in bit-reversed order). */
int nttx_dev_ntt_u32(void *d_x, int len, uint32_t p, int inverse, int bit_reversed_out, int batch, void *stream);

/* ---- building blocks for work split over several processes (one per GPU) --------------- */

/* The k-th modulus of the CRT path, k = 0..3 (the constants of main.cc:245-248). */
int nttx_crt_prime(int k, uint32_t *prime);

/* Which of those moduli nttx_poly_multiply_crt_u64 / nttx_dev_crt_multiply_u64 use for n coefficients modulo p:
 * main.cc:245-338 always takes all four; here as few of them as keep n*(p-1)^2 below their product (the
 * coefficients are the same).  Writes the moduli to primes[0..count-1] and returns count (1..4), or
 * NTTX_ERANGE when even four do not suffice.  Needs no device. */
int nttx_crt_primes_for(uint64_t p, int n, uint32_t primes[4]);

/* The per-prime part of the CRT path (residue split main.cc:292-303 and ntt_compute_omp
 * main.cc:306-311) for the primes selected by prime_mask (bit k = modulus k): d_res receives, for
 * each selected prime in ascending k, the [batch][len] canonical u32 residues of the products
 * (len = smallest power of two >= 2n).  d_a, d_b are [batch][n] u64 in device memory.  A process
 * that owns some of the primes calls this, exchanges residue slices with its peers and recombines
 * with nttx_dev_crt_recombine_u64 (the reference gives each prime a thread, main_pthread.cc:413-428). */
int nttx_dev_crt_residues_u64(const void *d_a, const void *d_b, void *d_res, int n, int prime_mask, int batch,
                              void *stream);

/* The same with the residues laid out for the exchange between the processes that share the primes: d_res is
 * [slice][selected primes][batch][len / slices], slice q holding indices q*len/slices .. (q+1)*len/slices - 1 of every
 * vector, i.e. the send buffer of an all-to-all over `slices` processes as it stands (no reordering copy; the last
 * inverse pass writes it).  slices a power of two; lengths 2^15 .. 2^20 (two-pass plans); slices = 1 is the call above. */
int nttx_dev_crt_residues_sliced_u64(const void *d_a, const void *d_b, void *d_res, int n, int prime_mask, int batch,
                                     int slices, void *stream);

/* CRT recombination of a slice (main.cc:323-337, crt_thread_func main_pthread.cc:293-310 works on
 * [start,end) the same way): d_out[i] = x_i mod p for i < count, where x_i < M has the residues
 * d_rk[i] modulo modulus k.  All device memory, u32 in, u64 out. */
int nttx_dev_crt_recombine_u64(const void *d_r0, const void *d_r1, const void *d_r2, const void *d_r3, void *d_out,
                               long long count, uint64_t p, void *stream);

/* One column pass of a length-2^log_len transform whose vector, seen as a 2^log_t x S matrix
 * (S = 2^(log_len-log_t), element (j, s) = v[j*S + s]), is split by columns over processes: d_x is
 * this process's dense [2^log_t][cols] slab of columns col0 .. col0+cols-1, transformed in place.
 * forward: length-2^log_t transform down every column (rows come out in bit-reversed order), then
 * element (j', s) times w^((col0+s) * bitrev(j')), w the primitive 2^log_len-th root
 * 3^((p-1)/2^log_len) of main.cc:156; inverse: the inverse factors and 2^-log_t first, then the
 * inverse transform taking rows in bit-reversed order.  zero_hi != 0 (forward): rows from
 * 2^(log_t-1) on are known to be zero and are not read.  Outputs canonical.  Together with a
 * transpose between the processes and length-S products of the rows this is one product of
 * length 2^log_len (no counterpart in the reference, whose arrays stop at 300000, main.cc:228).
 * Limits: 6 <= log_t <= 9, log_len <= 27, cols a power of two >= 16 that divides col0. */
int nttx_dev_colpass_u32(void *d_x, int log_len, int log_t, long long col0, int cols, uint32_t p, int inverse,
                         int zero_hi, void *stream);

/* Cyclic products of `rows` vectors of length len held in the layout an all-to-all between processes leaves them in:
 * d_a, d_b, d_c are [pieces][rows][len / pieces] u32 arrays, vector r being the concatenation of its pieces
 * (piece q = what process q sent, resp. will receive).  Same products as nttx_dev_cyclic_multiply_u32 on [rows][len]
 * arrays, without the two reordering copies around it: the first pass reads the pieces, the last one writes them.
 * len = 2^15 .. 2^17, pieces a power of two up to len / 512.  Part of the column-split product above; no counterpart
 * in the reference. */
int nttx_dev_cyclic_multiply_pieces_u32(const void *d_a, const void *d_b, void *d_c, int len, int pieces, int rows, uint32_t p,
                                        void *stream);

/* Pre-size the internal workspace so that later device-resident calls allocate nothing. */
int nttx_reserve(int len, int batch, int nprimes);
int nttx_reserve_u64(int len, int batch);   /* same for the 64-bit-lane direct path */

/* Per-kernel device timing with HIP events recorded on the launch stream around every
 * kernel of the path (no counterpart in the reference, which times the whole block with
 * std::chrono, main.cc:283,340).  Kinds: */
#define NTTX_K_COL_FWD 0 /* forward column pass (strided, four-step twiddle on the way out) */
#define NTTX_K_ROW_MUL 1 /* rows: forward a, forward b, pointwise, inverse */
#define NTTX_K_COL_INV 2 /* inverse column pass */
#define NTTX_K_ROW_FWD 3
#define NTTX_K_ROW_INV 4
#define NTTX_K_SPLIT 5   /* u64 -> residues */
#define NTTX_K_CRT 6     /* CRT recombination */
#define NTTX_K_MISC 7
#define NTTX_K_COUNT 8
int nttx_profile_enable(int on);
int nttx_profile_reset(void);
/* Waits for the recorded events, then returns accumulated milliseconds and launch count. */
int nttx_profile_read(int kind, double *total_ms, long long *launches);

/* Measurement aid (no counterpart in the reference): a plain streaming copy of `bytes` bytes on device memory by the
 * library's own 16-bytes-per-lane kernel, so that a benchmark can quote the passes of a product against the rate a
 * read-once / write-once kernel reaches on the same GPU.  Pointers 16-byte aligned, bytes a multiple of 16. */
int nttx_dev_stream_copy(void *d_dst, const void *d_src, size_t bytes, void *stream);

/* Decomposition chosen for a length: writes up to 3 pass sizes (log2) into logt, returns
 * the number of passes, or a negative code. */
int nttx_plan_info(int len, int *logt);
/* The same for the 64-bit-lane path (nttx_poly_multiply_direct_u64 and friends), where the longest row pass
 * depends on the headroom the modulus leaves in 64 bits. */
int nttx_plan_info_u64(int len, uint64_t p, int *logt);

#ifdef __cplusplus
}
#endif
#endif /* NTTX_H */

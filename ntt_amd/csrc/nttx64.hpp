// Internal interface between the C ABI (nttx.hip) and the 64-bit-lane direct path
// (nttx64.hip): plans, tables and launches for one transform modulo p itself when
// 2^31 <= p < 2^62 is an NTT-friendly prime.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nttx {

struct Engine64;   // plan cache + workspace of the 64-bit path, one per context

// Per-kernel timing hooks of the caller's profiler (kinds as in include/nttx.h).
struct ProfHook64 {
  void *self;
  void (*begin)(void *self, int kind, hipStream_t st);
  void (*end)(void *self, hipStream_t st);
};

Engine64 *engine64_create();
void engine64_destroy(Engine64 *e);
// message of the calling thread's last failure inside this unit
const char *engine64_error();

// p takes the 64-bit direct path for products of length 2^logL
bool engine64_ok(uint64_t p, int logL);
// passes chosen for a length (up to 3), returns the count
int engine64_decompose(int logL, uint64_t p, int lt[3], bool fp = false);   // the pass sizes depend on the modulus and the arithmetic class
// p runs its plain products on the floating-point class
bool engine64_fp_ok(uint64_t p);

// c = cyclic product of a and b, all [batch][2^logL] u64 device arrays, inputs below 2p, outputs
// canonical.  nvalid > 0: only the first nvalid words of each input vector are read.
int engine64_cyclic_multiply(Engine64 *e, const uint64_t *a, const uint64_t *b, uint64_t *c, int logL, uint64_t p,
                             int batch, long long nvalid, hipStream_t st, const ProfHook64 *prof);
// c = a b mod (x^len + 1, p), len = 2^logL, all [batch][len] u64 device arrays; inputs any 64-bit words, outputs
// canonical; needs 2^(logL+1) | p-1
int engine64_negacyclic_multiply(Engine64 *e, const uint64_t *a, const uint64_t *b, uint64_t *c, int logL, uint64_t p,
                                 int batch, hipStream_t st, const ProfHook64 *prof);
// in-place transform, natural in -> bit-reversed out, root w or (inverse != 0) w^-1, no scaling
int engine64_forward(Engine64 *e, uint64_t *x, int logL, uint64_t p, int inverse, int batch, hipStream_t st,
                     const ProfHook64 *prof);
// compact [batch][n] a and b -> zero-padded [batch][len] (values >= p reduced); zfill as in the 32-bit path
int engine64_pad(Engine64 *e, const uint64_t *a, const uint64_t *b, uint64_t *pa, uint64_t *pb, int n, int logL,
                 uint64_t p, int batch, hipStream_t st);
int engine64_unpad(const uint64_t *src, uint64_t *dst, int nout, int logL, int batch, hipStream_t st);
// workspace for later calls of this size (so that they allocate nothing)
int engine64_reserve(Engine64 *e, int logL, int batch);

}  // namespace nttx

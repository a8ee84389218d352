"""CPU stand-in for ntt_amd.distributed.HipEngine, built on the oracle: the same four operations on
CPU torch tensors, so the multi-rank logic (layouts, exchanges, slicing) can run under gloo or
in-process ranks without a GPU, and so GPU results of the individual operations have a checker.
Test infrastructure only."""
import numpy as np
import torch

CRT_PRIMES = (469762049, 998244353, 1004535809, 1224736769)   # main.cc:245-248


def bitrev_perm(bits):
    idx = np.arange(1 << bits)
    out = np.zeros_like(idx)
    for i in range(bits):
        out |= ((idx >> i) & 1) << (bits - 1 - i)
    return out


def _u32(t):
    return t.numpy().view(np.uint32)


def _u64(t):
    return t.numpy().view(np.uint64)


class OracleEngine:
    def __init__(self, oracle):
        self.O = oracle

    @staticmethod
    def product_len(n):
        k = 0
        while (1 << k) < 2 * n:
            k += 1
        return 1 << k

    def crt_residues(self, a, b, n, prime_mask):
        A, B = _u64(a), _u64(b)
        batch = A.shape[0]
        length = self.product_len(n)
        sel = [k for k in range(4) if prime_mask >> k & 1]
        res = np.zeros((len(sel), batch, length), dtype=np.uint32)
        for i, k in enumerate(sel):
            m = CRT_PRIMES[k]
            for r in range(batch):
                pa = np.zeros(length, dtype=np.uint32)
                pb = np.zeros(length, dtype=np.uint32)
                pa[:n] = (A[r] % np.uint64(m)).astype(np.uint32)
                pb[:n] = (B[r] % np.uint64(m)).astype(np.uint32)
                res[i, r] = self.O.ntt_compute(pa, pb, m)
        return torch.from_numpy(res.view(np.int32))

    def crt_residues_sliced(self, a, b, n, prime_mask, slices):
        length = self.product_len(n)
        if length < 64:                 # (a length without sliced kernels: the caller must take the other path)
            return None
        res = self.crt_residues(a, b, n, prime_mask)
        nsel, batch, _ = res.shape
        return res.view(nsel, batch, slices, length // slices).permute(2, 0, 1, 3).contiguous()

    def crt_recombine(self, residues, p):
        shape = residues[0].shape
        flat = [np.ascontiguousarray(_u32(r.contiguous()).reshape(-1)) for r in residues]
        out = self.O.crt_combine(flat, len(flat[0]), p)
        return torch.from_numpy(out.view(np.int64).reshape(tuple(shape)))

    def cyclic_multiply_pieces(self, a, b, p):
        """[pieces][rows][part] layout in and out: row r is the concatenation of its pieces."""
        pieces, rows, part = a.shape
        if pieces * part < 64:          # (a length without pieces-layout kernels: the caller must take the other path)
            return None
        ya = a.permute(1, 0, 2).reshape(rows, pieces * part).contiguous()
        yb = b.permute(1, 0, 2).reshape(rows, pieces * part).contiguous()
        yc = self.cyclic_multiply(ya, yb, p)
        return yc.view(rows, pieces, part).permute(1, 0, 2).contiguous()

    def colpass(self, x, log_len, log_t, col0, p, inverse, zero_hi=False):
        X = _u32(x)                      # view: written back in place
        T, cols = X.shape
        assert T == 1 << log_t
        L = 1 << log_len
        w = pow(3, (p - 1) // L, p)      # main.cc:156
        if inverse:
            w = pow(w, p - 2, p)         # main.cc:157
        br = bitrev_perm(log_t)
        # factor of element (j', s): w^((col0+s) * bitrev(j'))
        fac = np.empty((T, cols), dtype=np.uint64)
        for j in range(T):
            base = pow(w, int(br[j]), p)
            cur = pow(base, col0, p)
            for s in range(cols):
                fac[j, s] = cur
                cur = cur * base % p
        if not inverse:
            if zero_hi:
                assert not X[T // 2:].any()
            for s in range(cols):
                col = self.O.ntt_iter_barrett(np.ascontiguousarray(X[:, s]), p, False)
                X[:, s] = col[br]
            X[:] = (X.astype(np.uint64) * fac % np.uint64(p)).astype(np.uint32)
        else:
            tinv = pow(T, p - 2, p)
            Y = (X.astype(np.uint64) * fac % np.uint64(p) * np.uint64(tinv) % np.uint64(p)).astype(np.uint32)
            for s in range(cols):
                nat = np.empty(T, dtype=np.uint32)
                nat[br] = Y[:, s]
                X[:, s] = self.O.ntt_iter_barrett(nat, p, True)
        return x

    def cyclic_multiply(self, a, b, p):
        A, B = _u32(a), _u32(b)
        out = np.empty_like(A)
        for r in range(A.shape[0]):
            out[r] = self.O.ntt_compute(np.ascontiguousarray(A[r]), np.ascontiguousarray(B[r]), p)
        return torch.from_numpy(out.view(np.int32))

"""Products that span several GPUs: one process per GPU, `torch.distributed` (RCCL on the GPUs)
for the exchange steps.

Two workloads have a real exchange step:

* `CrtSharded` - the 4-prime CRT product with the primes dealt to ranks (the reference gives every
  prime a thread, main.cc:306-311 / main_pthread.cc:413-428, and cuts the recombination loop into
  index ranges, main_pthread.cc:435-465).  Each rank transforms for its own primes, one all-to-all
  hands every rank all four residues of its slice of the indices, and the recombination
  (main.cc:323-337) runs on slices.
* `FourStepSharded` - one product too long for a single pass structure to stay on one GPU: the
  vector is a T x S matrix split by columns; column passes are local, a transpose (all-to-all)
  turns column slabs into whole rows, the rows are multiplied locally, and the way back mirrors
  it: three all-to-alls per product, one per vector (a's is in flight while b's column pass runs).

Independent products need no exchange at all; they are sharded with `sharding.shard_range`.

The arithmetic is behind an `engine` object.  `HipEngine` (the product) calls the C ABI of
libnttx.so on device tensors and has no CPU fallback.  The communication is behind a `comm`
object: `TorchComm` wraps a torch.distributed process group, `ThreadComm` runs the ranks of a
group as threads of one process (for exercising the multi-rank logic on a single GPU).
Residues travel as int32 tensors and coefficients as int64 tensors holding the unsigned bit
patterns (torch has no arithmetic on unsigned 32/64-bit types, and none is needed here).
"""
from __future__ import annotations

import threading

from .sharding import shard_range

CRT_PRIME_COUNT = 4


# --------------------------------------------------------------------------------------------
# communication
# --------------------------------------------------------------------------------------------
class TorchComm:
    """A torch.distributed process group (backend "nccl" = RCCL on GPUs, "gloo" on CPU tensors)."""

    def __init__(self, group=None, ranks=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.ranks = list(ranks) if ranks is not None else list(range(dist.get_world_size()))
        self.world = len(self.ranks)
        self.rank = self.ranks.index(dist.get_rank())

    def all_to_all(self, send):
        """send[q] goes to rank q; returns recv with recv[q] = what rank q sent here."""
        import torch
        send = send.contiguous()
        if send.shape[0] != self.world:
            raise ValueError("leading dimension must equal the group size")
        recv = torch.empty_like(send)
        self._dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def all_to_all_start(self, send):
        """The same exchange, started and not waited for: returns a handle whose wait() gives recv.  Over RCCL the
        exchange runs on the communicator's own stream behind the work already enqueued on the current stream, so
        kernels launched after this call overlap with it."""
        import torch
        send = send.contiguous()
        if send.shape[0] != self.world:
            raise ValueError("leading dimension must equal the group size")
        recv = torch.empty_like(send)
        work = self._dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        return _Pending(work, recv, send)

    def all_gather(self, x):
        import torch
        x = x.contiguous()
        out = torch.empty((self.world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        self._dist.all_gather(list(out.unbind(0)), x, group=self.group)
        return out

    def barrier(self):
        self._dist.barrier(group=self.group)

    def split(self, color_of_rank):
        """Sub-communicators: ranks with equal color_of_rank(rank) form a group.  Collective over
        the whole communicator (every rank creates every group, as torch.distributed requires)."""
        colors = {}
        for r in range(self.world):
            colors.setdefault(color_of_rank(r), []).append(r)
        mine = None
        for color in sorted(colors):
            members = colors[color]
            g = self._dist.new_group([self.ranks[m] for m in members])
            if self.rank in members:
                mine = TorchComm(g, [self.ranks[m] for m in members])
        return mine


class _Pending:
    """An exchange in flight: wait() blocks the current stream (not the host, over RCCL) until recv is complete."""

    def __init__(self, work, recv, keep=None):
        self._work, self._recv, self._keep = work, recv, keep      # keep: the send buffer must outlive the exchange

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._recv


class _ThreadWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadComm:
    """Ranks of one process: `ThreadComm.run(world, fn)` starts fn(comm) on `world` threads.
    Collectives exchange tensor references through shared slots between two barriers."""

    def __init__(self, shared, rank):
        self._w = shared
        self.world = shared.world
        self.rank = rank

    @staticmethod
    def run(world, fn):
        shared = _ThreadWorld(world)
        results, errors = [None] * world, [None] * world

        def body(r):
            try:
                results[r] = fn(ThreadComm(shared, r))
            except BaseException as e:  # noqa: BLE001 - reported to the caller below
                errors[r] = e
                shared.barrier.abort()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errors:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in errors:
            if e is not None:
                raise e
        return results

    def _exchange(self, x):
        w = self._w
        w.slots[self.rank] = x
        w.barrier.wait()
        seen = list(w.slots)
        w.barrier.wait()
        return seen

    def all_to_all(self, send):
        import torch
        if send.shape[0] != self.world:
            raise ValueError("leading dimension must equal the group size")
        if send.is_cuda:
            torch.cuda.current_stream().synchronize()
        seen = self._exchange(send.contiguous())
        return torch.stack([seen[q][self.rank] for q in range(self.world)])

    def all_to_all_start(self, send):
        return _Pending(None, self.all_to_all(send))

    def all_gather(self, x):
        import torch
        if x.is_cuda:
            torch.cuda.current_stream().synchronize()
        return torch.stack(self._exchange(x.contiguous()))

    def barrier(self):
        self._w.barrier.wait()

    def split(self, color_of_rank):
        colors = {}
        for r in range(self.world):
            colors.setdefault(color_of_rank(r), []).append(r)
        # the first member of every group creates the shared state, everyone picks it up
        created = {}
        for color, members in colors.items():
            if members[0] == self.rank:
                created[color] = _ThreadWorld(len(members))
        seen = self._exchange(created)
        my_color = color_of_rank(self.rank)
        members = colors[my_color]
        return ThreadComm(seen[members[0]][my_color], members.index(self.rank))


# --------------------------------------------------------------------------------------------
# arithmetic on the GPU (C ABI of libnttx.so)
# --------------------------------------------------------------------------------------------
class HipEngine:
    """Device-tensor front of the C ABI.  Every call raises if the library, the GPU or the
    tensors' placement is not what the kernels need; there is no other code path."""

    def __init__(self):
        import torch
        from . import api
        self._t = torch
        self._api = api
        if not torch.cuda.is_available():
            raise api.NttxError(-3, "HipEngine needs a GPU")

    def _stream(self):
        return self._t.cuda.current_stream().cuda_stream

    def _chk(self, x, dtype):
        if not (x.is_cuda and x.is_contiguous() and x.dtype == dtype):
            raise ValueError(f"expected a contiguous device tensor of {dtype}")

    @staticmethod
    def product_len(n):
        k = 0
        while (1 << k) < 2 * n:
            k += 1
        return 1 << k

    def crt_residues(self, a, b, n, prime_mask):
        """a, b: [batch][n] int64 -> [primes in mask][batch][len] int32."""
        t = self._t
        self._chk(a, t.int64)
        self._chk(b, t.int64)
        batch = a.shape[0]
        nsel = bin(prime_mask).count("1")
        res = t.empty((nsel, batch, self.product_len(n)), dtype=t.int32, device=a.device)
        self._api.dev_crt_residues_u64(a.data_ptr(), b.data_ptr(), res.data_ptr(), n, prime_mask, batch, self._stream())
        return res

    def crt_multiply(self, a, b, n, p):
        """a, b: [batch][n] int64 -> [batch][2n-1] int64: the whole four-prime product on one GPU in one call (residue
        split, products, recombination: main.cc:286-338)."""
        t = self._t
        self._chk(a, t.int64)
        self._chk(b, t.int64)
        out = t.empty((a.shape[0], 2 * n - 1), dtype=t.int64, device=a.device)
        self._api.dev_crt_multiply_u64(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, a.shape[0], self._stream())
        return out

    def crt_residues_sliced(self, a, b, n, prime_mask, slices):
        """The same residues as [slices][primes in mask][batch][len / slices] (slice q = indices q*len/slices ..): the
        send buffer of the exchange between the processes that share the primes, written in place by the last pass.
        None when the length has no such kernels (the caller then slices the ordinary layout with a copy)."""
        t = self._t
        self._chk(a, t.int64)
        self._chk(b, t.int64)
        length = self.product_len(n)
        if not (1 << 15 <= length <= 1 << 20):
            return None
        batch = a.shape[0]
        nsel = bin(prime_mask).count("1")
        res = t.empty((slices, nsel, batch, length // slices), dtype=t.int32, device=a.device)
        self._api.dev_crt_residues_sliced_u64(a.data_ptr(), b.data_ptr(), res.data_ptr(), n, prime_mask, batch, slices, self._stream())
        return res

    def crt_recombine(self, residues, p):
        """residues: four int32 tensors of equal shape -> int64 tensor of that shape."""
        t = self._t
        for r in residues:
            self._chk(r, t.int32)
        out = t.empty(residues[0].shape, dtype=t.int64, device=residues[0].device)
        self._api.dev_crt_recombine_u64([r.data_ptr() for r in residues], out.data_ptr(), residues[0].numel(), p, self._stream())
        return out

    def colpass(self, x, log_len, log_t, col0, p, inverse, zero_hi=False):
        """In place on this rank's [2^log_t][cols] int32 slab."""
        self._chk(x, self._t.int32)
        if x.shape[0] != (1 << log_t):
            raise ValueError("slab must have 2^log_t rows")
        self._api.dev_colpass_u32(x.data_ptr(), log_len, log_t, col0, x.shape[1], p, inverse, zero_hi, self._stream())
        return x

    def cyclic_multiply(self, a, b, p):
        """Row-wise cyclic products of two [rows][len] int32 tensors."""
        t = self._t
        self._chk(a, t.int32)
        self._chk(b, t.int32)
        c = t.empty_like(a)
        self._api.dev_cyclic_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), a.shape[1], p, a.shape[0], self._stream())
        return c


    def cyclic_multiply_pieces(self, a, b, p):
        """Row-wise cyclic products of two [pieces][rows][len / pieces] int32 tensors (row r = the concatenation of
        its pieces, as an all-to-all delivers it); the result comes in the same layout.  None when the length has no
        pieces-layout kernels (the caller then reorders and uses cyclic_multiply)."""
        t = self._t
        self._chk(a, t.int32)
        self._chk(b, t.int32)
        pieces, rows, part = a.shape
        length = pieces * part
        if not (1 << 15 <= length <= 1 << 17) or pieces > length // 512:
            return None
        c = t.empty_like(a)
        self._api.dev_cyclic_multiply_pieces_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), length, pieces, rows, p, self._stream())
        return c


# --------------------------------------------------------------------------------------------
# 4-prime CRT products, primes dealt to ranks
# --------------------------------------------------------------------------------------------
def crt_layout(world):
    """(prime groups, batch groups): the primes are dealt over min(world, 4) ranks when that
    divides evenly, and whole copies of that arrangement share the batch."""
    for gp in (4, 2, 1):
        if world % gp == 0:
            return gp, world // gp
    return 1, world


class CrtSharded:
    """rank r: prime group r % Gp owns the moduli k with k % Gp == r % Gp, batch group r // Gp owns
    a contiguous share of the products.  The ranks of a batch group exchange residue slices once."""

    def __init__(self, comm, engine, force_exchange=False):
        """force_exchange: take the exchange route even where one rank owns every prime (the sliced send buffer, the
        all-to-all and the all-gather then run over a group of one: a way to drive the multi-rank code path, collective
        library included, on a single GPU)."""
        self.comm, self.engine = comm, engine
        self.force_exchange = force_exchange
        self.gp, self.gb = crt_layout(comm.world)
        gp = self.gp
        self.pr, self.br = comm.rank % gp, comm.rank // gp
        self.own = [k for k in range(CRT_PRIME_COUNT) if k % gp == self.pr]
        self.mask = sum(1 << k for k in self.own)
        self.group = comm.split(lambda r: r // gp) if comm.world > 1 else comm

    def batch_range(self, batch):
        """Products [lo, hi) whose inputs this rank needs (the same for every rank of its batch group)."""
        return shard_range(batch, self.br, self.gb)

    def multiply(self, a, b, n, p, gather=True):
        """a, b: [nb][n] int64, this batch group's products.  Returns [nb][2n-1] int64 when gather,
        else this rank's slice [nb][len/Gp] of the coefficient indices (slice q = indices
        q*len/Gp .. (q+1)*len/Gp - 1, of which only those below 2n-1 are meaningful)."""
        gp = self.gp
        # all four moduli are in play here whatever p is; the product is exact only while n*(p-1)^2 stays below their
        # product ("sum % M", main.cc:330) - the same check, and the same NTTX_ERANGE, as the single-GPU entry point
        from . import api
        api.crt_primes_for(p, n)
        if gp == 1 and not self.force_exchange:
            if hasattr(self.engine, "crt_multiply"):
                # one rank owns every prime and the whole index range: the single-GPU entry point does all of it in one
                # call (the first column pass reads the u64 coefficients itself, the residues never leave the library's
                # workspace); [nb][2n-1], the meaningful indices, with or without gather
                return self.engine.crt_multiply(a, b, n, p)
            res = self.engine.crt_residues(a, b, n, self.mask)          # [nown][nb][len]
            full = self.engine.crt_recombine([res[k] for k in range(CRT_PRIME_COUNT)], p)
            return full[:, : 2 * n - 1] if gather else full
        # [dest][nown][nb][sl]: written in that order by the last inverse pass where the length allows it, otherwise
        # sliced out of the ordinary layout with one copy
        send = self.engine.crt_residues_sliced(a, b, n, self.mask, gp)
        if send is None:
            res = self.engine.crt_residues(a, b, n, self.mask)          # [nown][nb][len]
            nown, nb, length = res.shape
            send = res.view(nown, nb, gp, length // gp).permute(2, 0, 1, 3).contiguous()
        _, nown, nb, sl = send.shape
        length = sl * gp
        recv = self.group.all_to_all(send)                                    # [source][nown][nb][sl]
        by_prime = [recv[k % gp, k // gp] for k in range(CRT_PRIME_COUNT)]
        part = self.engine.crt_recombine(by_prime, p)                          # [nb][sl]
        if not gather:
            return part
        parts = self.group.all_gather(part)                                    # [gp][nb][sl]
        return parts.permute(1, 0, 2).reshape(nb, length)[:, : 2 * n - 1]


# --------------------------------------------------------------------------------------------
# one long product split by columns
# --------------------------------------------------------------------------------------------
class FourStepSharded:
    """Cyclic product of two length-2^log_len vectors over `comm.world` = G ranks (a power of two).

    Layout: element i = j*S + s of a vector sits at (row j, column s) of a T x S matrix
    (T = 2^log_t); rank g holds columns g*S/G .. (g+1)*S/G - 1 as a dense [T][S/G] int32 slab, for
    inputs and for the result alike.  `slab_of` / `place_slab` convert from and to whole vectors."""

    def __init__(self, comm, engine, log_len, p, log_t=None, force_exchange=False):
        """force_exchange: a single rank also goes through the three exchanges and the pieces-layout row products (over a
        group of one), so the multi-rank code path can be driven on a single GPU."""
        self.force_exchange = force_exchange
        g = comm.world
        if g & (g - 1):
            raise ValueError("the number of ranks must be a power of two")
        if log_t is None:
            log_t = min(9, log_len // 2)
        self.comm, self.engine, self.p = comm, engine, p
        self.log_len, self.log_t = log_len, log_t
        self.T, self.S = 1 << log_t, 1 << (log_len - log_t)
        if self.T % g or self.S % g or self.S // g < 16:
            raise ValueError("need G | T and at least 16 columns per rank")
        self.cols = self.S // g
        self.col0 = comm.rank * self.cols
        self.rows = self.T // g

    def slab_of(self, vec):
        """This rank's slab of a whole length-2^log_len vector (testing / data loading)."""
        return vec.view(self.T, self.S)[:, self.col0: self.col0 + self.cols].contiguous()

    def place_slab(self, vec, slab):
        vec.view(self.T, self.S)[:, self.col0: self.col0 + self.cols] = slab

    def _send_rows(self, x):
        """Start the exchange that turns this rank's column slab of one vector into whole rows.  The slab is row-major,
        so its [dest][rows][cols] view is the send buffer as it stands: no packing copy."""
        return self.comm.all_to_all_start(x.view(self.comm.world, self.rows, self.cols))

    @staticmethod
    def _rows_of(recv):
        g, rows, cols = recv.shape                                               # [source][rows][cols]
        return recv.permute(1, 0, 2).reshape(rows, g * cols)                     # (one copy: rows become contiguous)

    def _to_slab(self, y):
        g = self.comm.world
        if g == 1 and not self.force_exchange:
            return y
        z = y.view(self.rows, g, self.cols).permute(1, 0, 2).contiguous()    # [dest][rows][cols]
        return self.comm.all_to_all(z).view(self.T, self.cols)

    def multiply(self, xa, xb, zero_hi=False):
        """xa, xb: this rank's slabs (overwritten).  zero_hi: both inputs are zero from index
        2^(log_len-1) on (a linear product), so the upper half of the rows is not read.
        Returns this rank's slab of the product."""
        e, L, t, p = self.engine, self.log_len, self.log_t, self.p
        if self.comm.world == 1 and not self.force_exchange:
            e.colpass(xa, L, t, self.col0, p, False, zero_hi)
            e.colpass(xb, L, t, self.col0, p, False, zero_hi)
            ya, yb = xa, xb
        else:
            # a's exchange is in flight while b's column pass runs
            e.colpass(xa, L, t, self.col0, p, False, zero_hi)
            pa = self._send_rows(xa)
            e.colpass(xb, L, t, self.col0, p, False, zero_hi)
            pb = self._send_rows(xb)
            ra, rb = pa.wait(), pb.wait()                                        # [source][rows][cols] each
            # the rows are multiplied where the exchange left them and the products come out in the layout the way
            # back sends ([dest][rows][cols]): no reordering copy on either side
            zc = e.cyclic_multiply_pieces(ra, rb, p)
            if zc is not None:
                xc = self.comm.all_to_all(zc).view(self.T, self.cols)
                e.colpass(xc, L, t, self.col0, p, True)
                return xc
            ya, yb = self._rows_of(ra), self._rows_of(rb)
        yc = e.cyclic_multiply(ya.contiguous(), yb.contiguous(), p)
        xc = self._to_slab(yc).contiguous()
        e.colpass(xc, L, t, self.col0, p, True)
        return xc

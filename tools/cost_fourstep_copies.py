"""What one rank of the column-split product (ntt_amd.distributed.FourStepSharded) spends outside the exchange itself,
measured on ONE GPU with the shapes a rank would hold at G ranks: the column passes, the row products, and the
layout copies that a reordering implementation needs around the all-to-alls (stack / permute / contiguous), against
the row products computed directly in the exchanged layout (nttx_dev_cyclic_multiply_pieces_u32: no copies).  The all-to-all is replaced by an identity
(timing only, results are meaningless).  usage: cost_fourstep_copies.py [log2_len] [G]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntt_amd
from ntt_amd.distributed import FourStepSharded, HipEngine

logL = int(sys.argv[1]) if len(sys.argv) > 1 else 24
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
P = 469762049


class FakeComm:
    def __init__(self, world): self.world, self.rank = world, 0
    def all_to_all(self, send): return send.clone() if False else send      # identity: no time charged to the exchange
    def barrier(self): pass


ntt_amd.init(0)
fs = FourStepSharded(FakeComm(G), HipEngine(), logL, P)
dev = torch.device("cuda", 0)
xa = torch.randint(0, P, (fs.T, fs.cols), device=dev, dtype=torch.int64).to(torch.int32)
xb = torch.randint(0, P, (fs.T, fs.cols), device=dev, dtype=torch.int64).to(torch.int32)
xa[fs.T // 2:] = 0; xb[fs.T // 2:] = 0
e = fs.engine


def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3     # us


g = G
out = {"log2_len": logL, "ranks": G, "slab_words": fs.T * fs.cols, "rows_per_rank": fs.rows, "row_length": fs.S}
out["colpass_fwd_a_and_b_us"] = timeit(lambda: (e.colpass(xa, logL, fs.log_t, 0, P, False, True), e.colpass(xb, logL, fs.log_t, 0, P, False, True)))
send = None
def pack():
    global send
    send = torch.stack([x.view(g, fs.rows, fs.cols) for x in (xa, xb)], dim=1)
out["pack_for_exchange_us"] = timeit(pack)
recv = send
rows = None
def unpack():
    global rows
    rows = [recv[:, v].permute(1, 0, 2).reshape(fs.rows, fs.S) for v in range(2)]
out["unpack_to_rows_us"] = timeit(unpack)
ya, yb = rows[0].contiguous(), rows[1].contiguous()
yc = None
def rowprod():
    global yc
    yc = e.cyclic_multiply(ya, yb, P)
out["row_products_us"] = timeit(rowprod)
z = None
def pack2():
    global z
    z = yc.view(fs.rows, g, fs.cols).permute(1, 0, 2).contiguous()
out["pack_back_us"] = timeit(pack2)
xc = z.view(fs.T, fs.cols)
out["colpass_inv_us"] = timeit(lambda: e.colpass(xc, logL, fs.log_t, 0, P, True))
ra = xa.view(g, fs.rows, fs.cols).clone()
rb = xb.view(g, fs.rows, fs.cols).clone()
if e.cyclic_multiply_pieces(ra, rb, P) is not None:
    out["row_products_in_pieces_layout_us"] = timeit(lambda: e.cyclic_multiply_pieces(ra, rb, P))
out["copies_us"] = out["pack_for_exchange_us"] + out["unpack_to_rows_us"] + out["pack_back_us"]
out["kernels_us"] = out["colpass_fwd_a_and_b_us"] + out["row_products_us"] + out["colpass_inv_us"]
out["exchange_bytes_per_rank"] = int(3 * fs.T * fs.cols * 4 * (G - 1) / G)
print(json.dumps(out))
ntt_amd.shutdown()

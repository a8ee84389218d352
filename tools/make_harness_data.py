"""Write <dir>/{0..4}.in and .out in the course harness's text format (main.cc:78-128): id 0 is the reference's own
case, ids 1-4 seeded inputs of the harness sizes with the oracle's products as expected outputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po   # noqa: E402

P_ID = {1: 7340033, 2: 104857601, 3: 469762049, 4: 1337006139375617}   # test.o:3-11
d = sys.argv[1]
os.makedirs(d, exist_ok=True)
O = po.Oracle()
cases = [([1, 5, 5, 4], [4, 1, 5, 2], 7340033)]
for tid, p in P_ID.items():
    cases.append((O.fill_random(131072, 1000 + tid, p), O.fill_random(131072, 2000 + tid, p), p))
for i, (a, b, p) in enumerate(cases):
    po.write_in_file(os.path.join(d, f"{i}.in"), a, b, p)
    po.write_out_file(os.path.join(d, f"{i}.out"), O.poly_multiply_crt(a, b, p))
print("wrote", len(cases), "cases to", d)

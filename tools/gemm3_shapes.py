"""k_gemm3 (flat and super-tile order) beside k_gemm2 on the batched shapes of the C3 step (sc_dbg_gemm3_bench).

    python tools/gemm3_shapes.py [count]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from springcraft_amd import _hip  # noqa: E402

L = _hip.lib()
ctx = _hip.context()
fn = L.sc_dbg_gemm3_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.POINTER(C.c_double)]
count = int(sys.argv[1]) if len(sys.argv) > 1 else 32
print(ctx.info(), f"count {count}", flush=True)
shapes = [
    ("trailing update (NT lower)", 5936, 5936, 256, 2, 1),
    ("trailing update (NT lower)", 3056, 3056, 256, 2, 1),
    ("trailing update (NT lower)", 5936, 5936, 128, 2, 1),
    ("Q1 update (NN)", 5936, 6000, 256, 0, 0),
    ("Q1 update (NN)", 2864, 6000, 256, 0, 0),
    ("square (NN)", 6144, 6144, 256, 0, 0),
    ("square (NT)", 6144, 6144, 256, 2, 0),
]
if len(sys.argv) > 2 and sys.argv[2] == "dc":
    shapes = [("merge, top level (NN)", 3000, 3488, 3488, 0, 0), ("merge, level below (NN)", 1500, 1744, 1744, 0, 0),
              ("merge, third level (NN)", 750, 880, 880, 0, 0)]
if len(sys.argv) > 2 and sys.argv[2] == "ld":
    shapes = [("NN", 6000, 6000, 256, 0, 0), ("NT lower", 5936, 5936, 256, 2, 1)]
if len(sys.argv) > 2 and sys.argv[2] == "edges":
    shapes = [("NN", m, n, 256, 0, 0) for (m, n) in ((5888, 6016), (5936, 6016), (5888, 6000), (6016, 6016), (6144, 6000), (5936, 6144), (6000, 6000))]
    shapes += [("NT lower", m, m, 256, 2, 1) for m in (5888, 6016, 6144)] + [("NT full", m, m, 256, 2, 0) for m in (5888, 5936)]
for name, m, n, k, layout, lower in shapes:
    row = []
    flops = 2.0 * m * n * k * count * (0.5 if lower else 1.0)
    for label, kernel, order in (("k_gemm2", 2, 0), ("k_gemm3 flat", 3, 0), ("k_gemm3 super", 3, 1)):
        ms = C.c_double()
        rc = fn(ctx.handle, count, m, n, k, layout, lower, kernel, order, 4, C.byref(ms))
        tf = flops / ms.value / 1e9 if rc == 0 and ms.value > 0 else 0.0
        row.append(f"{label}: {ms.value:7.3f} ms {tf:5.1f} TF = {tf / 78.6:.3f} rc {rc}")
    print(f"{name:28s} {m:5d} x {n:5d} K {k:4d}  " + " | ".join(row), flush=True)

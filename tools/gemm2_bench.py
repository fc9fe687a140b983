"""Old grouped GEMM (64 x 64 x 8 tile) vs k_gemm2 on the shapes the eigensolver launches (sc_dbg_gemm_bench).

    python tools/gemm2_bench.py [quick]

tile ids: 3 = old kernel, 10 = k_gemm2 with the automatic block tile, 11 / 12 / 13 = 128x128 / 128x64 / 64x64 forced.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from springcraft_amd import _hip  # noqa: E402

L = _hip.lib()
ctx = _hip.context()
fn = L.sc_dbg_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_double)] * 2
print(ctx.info(), flush=True)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [
    # name, m, n, k, mode (0 NN, 1 NT lower, 2 TN), split, beta
    ("syr2k NT lower K=128", 6000, 6000, 128, 1, 1, 1),
    ("syr2k NT lower K=256", 6000, 6000, 256, 1, 1, 1),
    ("bt update NN K=128", 6000, 6000, 128, 0, 1, 1),
    ("bt update NN K=256", 6000, 6000, 256, 0, 1, 1),
    ("bt update NN K=64", 6000, 6000, 64, 0, 1, 1),
    ("symm NN 6000x64x6000", 6000, 64, 6000, 0, 1, 0),
    ("symm TN 6000x64x6000", 6000, 64, 6000, 2, 1, 0),
    ("W NN 6000x64x192", 6000, 64, 192, 0, 1, 0),
    ("gram TN 64x192x6000 s8", 64, 192, 6000, 2, 8, 0),
    ("bt W1 TN 128x6000x6000 s8", 128, 6000, 6000, 2, 8, 0),
    ("bt W1 TN 128x6000x6000 s2", 128, 6000, 6000, 2, 2, 0),
    ("square NN 6000", 6000, 6000, 6000, 0, 1, 0),
    ("square NT 3000", 3000, 3000, 3000, 1, 1, 0),
    ("square NN 1536", 1536, 1536, 1536, 0, 1, 0),
    ("ragged NN 1030x517x333", 1030, 517, 333, 0, 1, 1),
    ("ragged TN 257x1001x1999 s3", 257, 1001, 1999, 2, 3, 0),
    ("ragged NT 999x999x77", 999, 999, 77, 1, 1, 1),
]
if quick:
    shapes = shapes[:4] + shapes[-3:]
for name, m, n, k, mode, split, beta in shapes:
    row = []
    for tile in (3, 10, 11, 12, 13):
        ms = C.c_double()
        err = C.c_double()
        rc = fn(ctx.handle, m, n, k, mode, tile, split, 5, beta, C.byref(ms), C.byref(err))
        flops = 2.0 * m * n * k * (0.5 if mode == 1 else 1.0)
        row.append(f"t{tile}: {flops / ms.value / 1e9:6.2f} TF err {err.value:.1e} rc {rc}")
    print(f"{name:30s} " + " | ".join(row), flush=True)

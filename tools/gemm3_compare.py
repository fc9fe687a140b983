"""k_gemm2 (the library's kernel, sc_dbg_gemm_bench) on the shapes tools/probe_gemm3.hip times: eight 6144 x 6144 updates stacked
into one 49152 x 6144 product (the same number of tiles as the probe's batch of 8).   python tools/gemm3_compare.py"""
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
shapes = [("NN update 49152 x 6144", 49152, 6144, k, 0, 1) for k in (128, 256, 512)]
shapes += [("NT lower 24576^2", 24576, 24576, k, 1, 1) for k in (128, 256, 512)]
shapes += [("NN square 6144^3", 6144, 6144, 6144, 0, 0)]
for name, m, n, k, mode, beta in shapes:
    row = []
    for tile in (10, 11, 12):
        ms, err = C.c_double(), C.c_double()
        rc = fn(ctx.handle, m, n, k, mode, tile, 1, 5, beta, C.byref(ms), C.byref(err))
        flops = 2.0 * m * n * k * (0.5 if mode == 1 else 1.0)
        tf = flops / ms.value / 1e9
        row.append(f"t{tile}: {ms.value:8.3f} ms {tf:6.2f} TF = {tf / 78.6:.3f} rc {rc}")
    print(f"{name:26s} K = {k:5d}  " + " | ".join(row), flush=True)

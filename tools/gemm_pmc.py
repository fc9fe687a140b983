"""One GEMM shape per run for PMC collection: python tools/gemm_pmc.py m n k mode tile beta."""
import ctypes as C
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from springcraft_amd import _hip  # noqa: E402

m, n, k, mode, tile, beta = (int(x) for x in sys.argv[1:7])
L = _hip.lib()
ctx = _hip.context()
fn = L.sc_dbg_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_double)] * 2
ms = C.c_double()
err = C.c_double()
rc = fn(ctx.handle, m, n, k, mode, tile, 1, 3, beta, C.byref(ms), C.byref(err))
print(rc, ms.value, 2.0 * m * n * k * (0.5 if mode == 1 else 1.0) / ms.value / 1e9, "TFLOP/s")

"""GEMM shapes of the two-stage path (debug entry point sc_dbg_gemm_bench); batch emulated by a tall M."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from springcraft_amd import _hip  # noqa: E402

L = _hip.lib()
ctx = _hip.context()
fn = L.sc_dbg_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_double)] * 2
cases = [
    ("symm-like NN 48000x64x3000 t3", 48000, 64, 3000, 0, 3, 1, 0),
    ("symm-like TN 48000x64x3000 t3", 48000, 64, 3000, 2, 3, 1, 0),
    ("symm-like TN 48000x64x3000 t2", 48000, 64, 3000, 2, 2, 1, 0),
    ("symm-like NN 48000x64x3000 t2", 48000, 64, 3000, 0, 2, 1, 0),
    ("syr2k NT lower 12000^2 K=128 t3", 12000, 12000, 128, 1, 3, 1, 1),
    ("syr2k NT lower 12000^2 K=128 t2", 12000, 12000, 128, 1, 2, 1, 1),
    ("bt1 update NN 24000x6000 K=128 t3", 24000, 6000, 128, 0, 3, 1, 1),
    ("bt1 update NN 24000x6000 K=128 t2", 24000, 6000, 128, 0, 2, 1, 1),
    ("bt1 update NN 24000x6000 K=128 t0", 24000, 6000, 128, 0, 0, 1, 1),
    ("bt1 W1 TN 128x6000x24000 split8 t3", 128, 6000, 24000, 2, 3, 8, 0),
]
for name, m, n, k, mode, tile, split, beta in cases:
    ms = C.c_double()
    err = C.c_double()
    rc = fn(ctx.handle, m, n, k, mode, tile, split, 5, beta, C.byref(ms), C.byref(err))
    flops = 2.0 * m * n * k * (0.5 if mode == 1 else 1.0)
    print(f"{name:40s} rc={rc} {ms.value:9.3f} ms  {flops / ms.value / 1e9:8.2f} TFLOP/s  maxerr={err.value:.2e}", flush=True)

"""Times the f64 MFMA GEMM on the shapes the eigensolver uses (debug entry point sc_dbg_gemm_bench)."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from springcraft_amd import _hip  # noqa: E402

L = _hip.lib()
ctx = _hip.context()
fn = L.sc_dbg_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_double)] * 2
print(ctx.info())
cases = [
    # name, m, n, k, mode, tile, split, beta
    ("merge NN 6000^3", 6000, 6000, 6000, 0, 0, 1, 0),
    ("merge NN 3000^3", 3000, 3000, 3000, 0, 0, 1, 0),
    ("syr2k NT lower K=128", 6000, 6000, 128, 1, 0, 1, 1),
    ("syr2k NT lower K=256", 6000, 6000, 256, 1, 0, 1, 1),
    ("syr2k NT lower K=128 t2", 6000, 6000, 128, 1, 2, 1, 1),
    ("syr2k NT lower K=256 t2", 6000, 6000, 256, 1, 2, 1, 1),
    ("bt update NN K=128 t2", 6000, 6000, 128, 0, 2, 1, 1),
    ("bt update NN K=256 t2", 6000, 6000, 256, 0, 2, 1, 1),
    ("merge NN 3000^3 t2", 3000, 3000, 3000, 0, 2, 1, 0),
    ("merge NN 6000^3 t2", 6000, 6000, 6000, 0, 2, 1, 0),
    ("syr2k NT lower K=128 t3", 6000, 6000, 128, 1, 3, 1, 1),
    ("syr2k NT lower K=256 t3", 6000, 6000, 256, 1, 3, 1, 1),
    ("bt update NN K=128 t3", 6000, 6000, 128, 0, 3, 1, 1),
    ("merge NN 3000^3 t3", 3000, 3000, 3000, 0, 3, 1, 0),
    ("merge NN 6000^3 t3", 6000, 6000, 6000, 0, 3, 1, 0),
    ("bt W1 TN m=128 split8 t3", 128, 6000, 6000, 2, 3, 8, 0),
    ("bt W1 TN m=128 split4 t3", 128, 6000, 6000, 2, 3, 4, 0),
    ("bt W1 TN m=128 split2 t3", 128, 6000, 6000, 2, 3, 2, 0),
    ("bt update NN K=256 t3", 6000, 6000, 256, 0, 3, 1, 1),
    ("vt NN 6000x128x128 t3", 6000, 128, 128, 0, 3, 1, 0),
    ("square NN 1536 t3", 1536, 1536, 1536, 0, 3, 1, 0),
    ("bt W1 TN m=128 split8 t2", 128, 6000, 6000, 2, 2, 8, 0),
    ("bt W1 TN m=128 split4 t2", 128, 6000, 6000, 2, 2, 4, 0),
    ("bt update NN K=64", 6000, 6000, 64, 0, 0, 1, 1),
    ("bt update NN K=128", 6000, 6000, 128, 0, 0, 1, 1),
    ("bt update NN K=256", 6000, 6000, 256, 0, 0, 1, 1),
    ("bt W1 TN m=64 split8", 64, 6000, 6000, 2, 1, 8, 0),
    ("bt W1 TN m=128 split8", 128, 6000, 6000, 2, 0, 8, 0),
    ("bt W1 TN m=128 split16", 128, 6000, 6000, 2, 0, 16, 0),
    ("square NN 1536", 1536, 1536, 1536, 0, 0, 1, 0),
]
for name, m, n, k, mode, tile, split, beta in cases:
    ms = C.c_double()
    err = C.c_double()
    rc = fn(ctx.handle, m, n, k, mode, tile, split, 5, beta, C.byref(ms), C.byref(err))
    flops = 2.0 * m * n * k * (0.5 if mode == 1 else 1.0)
    print(f"{name:28s} rc={rc} {ms.value:9.3f} ms  {flops / ms.value / 1e9:8.2f} TFLOP/s  maxerr={err.value:.2e}", flush=True)

import ctypes as C, sys
sys.path.insert(0, ".")
from springcraft_amd import _hip
L = _hip.lib(); ctx = _hip.context()
fn = L.sc_dbg_gemm_bench; fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_double)] * 2
for name, m, n, k, mode, tile, split, beta in [
    ("NN 6000^3 t3", 6000, 6000, 6000, 0, 3, 1, 0),
    ("NN 6000x6000x1024 t3", 6000, 6000, 1024, 0, 3, 1, 0),
    ("NN 6000x6000x512 t3 b1", 6000, 6000, 512, 0, 3, 1, 1),
    ("NN 6000x6000x256 t3 b1", 6000, 6000, 256, 0, 3, 1, 1),
    ("NN 6000x6000x128 t3 b1", 6000, 6000, 128, 0, 3, 1, 1),
    ("NN 6000x6000x128 t3 b0", 6000, 6000, 128, 0, 3, 1, 0),
    ("NN 12000x12000x128 t3 b1", 12000, 12000, 128, 0, 3, 1, 1),
    ("NN 24000x6000x128 t3 b0", 24000, 6000, 128, 0, 3, 1, 0),
]:
    ms = C.c_double(); err = C.c_double()
    rc = fn(ctx.handle, m, n, k, mode, tile, split, 5, beta, C.byref(ms), C.byref(err))
    print(f"{name:32s} rc={rc} {ms.value:9.3f} ms  {2.0*m*n*k / ms.value / 1e9:8.2f} TFLOP/s", flush=True)

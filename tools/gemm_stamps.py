"""Diagnostic build only (SC_EXTRA_HIPCC_FLAGS=-DGEMM_STAMPS): mean shader cycles per wave of k_gemm2's prologue, K loop
and epilogue for a few shapes (batch emulated by a tall M where the shape allows).  python tools/gemm_stamps.py"""
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
st = L.sc_dbg_gemm_stamps
st.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 4)()
for name, m, n, k, mode, tile, beta in [
    ("syr2k lower 24000x24000 K=128 t11", 24000, 24000, 128, 1, 11, 1),
    ("syr2k lower 24000x24000 K=128 t13", 24000, 24000, 128, 1, 13, 1),
    ("update NN 48000x6000 K=256 t11", 48000, 6000, 256, 0, 11, 1),
    ("square NN 6000 t11", 6000, 6000, 6000, 0, 11, 0),
]:
    st(buf, 1)
    ms = C.c_double()
    err = C.c_double()
    rc = fn(ctx.handle, m, n, k, mode, tile, 1, 3, beta, C.byref(ms), C.byref(err))
    st(buf, 1)
    cnt = max(1, buf[3])
    flops = 2.0 * m * n * k * (0.5 if mode == 1 else 1.0)
    print(f"{name:38s} rc {rc} {flops / ms.value / 1e9:6.1f} TF  waves {cnt:8d}  prologue {buf[0] / cnt:9.0f}  K loop {buf[1] / cnt:9.0f}  "
          f"epilogue {buf[2] / cnt:8.0f} cyc/wave", flush=True)

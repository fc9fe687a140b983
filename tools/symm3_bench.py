"""k_symm3 alone on the shapes of the band reduction: python tools/symm3_bench.py [count m split]...  (TFLOP/s on 2 m^2 64 flops)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from springcraft_amd import _hip  # noqa: E402

L = _hip.lib()
ctx = _hip.context()
fn = L.sc_dbg_symm3_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
shapes = [(32, 5888, 1), (32, 3008, 1), (64, 5888, 1), (1, 23936, 9), (6, 5888, 6)]
args = [int(x) for x in sys.argv[1:]]
if args:
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)]
for count, m, split in shapes:
    ms = C.c_double(0.0)
    rc = fn(ctx.handle, count, m, split, 5, C.byref(ms))
    fl = 2.0 * m * m * 64 * count
    print(f"rc {rc}  {count} x m = {m}, {split} slices: {ms.value:.3f} ms  {fl / ms.value / 1e9:.1f} TFLOP/s = {fl / ms.value / 1e9 / 78.6:.3f} of the f64 MFMA peak"
          f"  [{os.environ.get('SPRINGCRAFT_SYMM3_DBG_HOT', '0')}]")

"""Diagnostic build only (SC_EXTRA_HIPCC_FLAGS=-DGEMM_STAMPS): per-workgroup timeline of k_gemm2 -- do the K loops of the
workgroups that share a CU overlap the C traffic (prologue / epilogue) of their neighbours?  python tools/gemm_trace.py"""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from springcraft_amd import _hip  # noqa: E402

L = _hip.lib()
ctx = _hip.context()
fn = L.sc_dbg_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_double)] * 2
tr = L.sc_dbg_gemm_trace
tr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
MAXR = 1 << 16
buf = np.zeros((MAXR, 6), dtype=np.uint64)
cnt = C.c_int()


def covered(intervals, lo, hi, depth):
    """time in [lo, hi) covered by at least `depth` of the intervals"""
    ev = []
    for a, b in intervals:
        ev.append((a, 1)); ev.append((b, -1))
    ev.sort()
    d, last, tot = 0, lo, 0
    for t, s in ev:
        if d >= depth:
            tot += t - last
        last = t
        d += s
    return tot


for name, m, n, k, mode, tile, beta in [
    ("syr2k lower 24000^2 K=128 128x128", 24000, 24000, 128, 1, 11, 1),
    ("update NN 48000x6000 K=256 128x128", 48000, 6000, 256, 0, 11, 1),
]:
    tr(None, 0, C.byref(cnt), 1)
    ms = C.c_double(); err = C.c_double()
    rc = fn(ctx.handle, m, n, k, mode, tile, 1, 1, beta, C.byref(ms), C.byref(err))
    tr(buf.ctypes.data_as(C.c_void_p), MAXR, C.byref(cnt), 1)
    R = buf[:cnt.value].astype(np.int64)
    flops = 2.0 * m * n * k * (0.5 if mode == 1 else 1.0)
    print(f"== {name}: rc {rc} {flops / ms.value / 1e9:.1f} TF, {cnt.value} workgroups recorded")
    hw = R[:, 0]
    key = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    cus = defaultdict(list)
    for i in range(len(R)):
        cus[int(key[i])].append(R[i])
    f0 = f1 = f2 = span_sum = 0.0
    for kcu, rows in cus.items():
        rows.sort(key=lambda r: r[1])
        lo = min(r[1] for r in rows); hi = max(r[4] for r in rows)
        loops = [(r[2], r[3]) for r in rows]
        c1 = covered(loops, lo, hi, 1); c2 = covered(loops, lo, hi, 2)
        span_sum += hi - lo
        f0 += (hi - lo) - c1; f1 += c1 - c2; f2 += c2
    print(f"   {len(cus)} CUs; share of a CU's time with 0 / 1 / >=2 workgroups inside their K loop: "
          f"{f0 / span_sum:.2f} / {f1 / span_sum:.2f} / {f2 / span_sum:.2f}")
    kcu = sorted(cus)[len(cus) // 2]
    rows = cus[kcu]
    t0 = rows[0][1]
    print(f"   CU {kcu:#x}: start / K loop / epilogue / end (cycles since the CU's first start), first 14 workgroups")
    for r in rows[:14]:
        print(f"      wg {r[5]:7d}  simd-wave {r[0] & 0x3f:#04x}  {r[1] - t0:8d} {r[2] - t0:8d} {r[3] - t0:8d} {r[4] - t0:8d}")

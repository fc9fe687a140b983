"""k_sytrd_resident (one launch, the matrix resident in LDS) against the launches per column and against LAPACK:
accuracy, wall-clock of nma.eigh (host API, one matrix), the take-over routes, the number of workgroups.

    python tools/resident_check.py [sizes ...]          e.g. 300 900 1536 2048 3000
    python tools/resident_check.py --wgs 1536           sweep of the workgroup count at one size
"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from springcraft_amd import nma, _hip  # noqa: E402

L = _hip.lib()
ctx = _hip.context()
L.sc_dbg_set_resident.restype = C.c_int
L.sc_dbg_set_resident.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
L.sc_last_eigh_phase_ms.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]


def counter(name):
    v = C.c_int64(0)
    ctx.check(L.sc_ctx_get_counter(ctx.handle, name.encode(), C.byref(v)))
    return v.value


def run(a, mode, hook=0, wgs=0, reps=3, vectors=True):
    ctx.check(L.sc_dbg_set_resident(ctx.handle, mode, hook, wgs))
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = nma.eigh(a) if vectors else nma.eigh(a, eigenvectors=False)
        best = min(best, time.perf_counter() - t0)
    ctx.check(L.sc_dbg_set_resident(ctx.handle, -1, 0, 0))
    return out, best


def errors(a, w, v):
    n = len(a)
    wr = np.linalg.eigvalsh(a)
    scale = max(np.abs(wr).max(), 1e-300)
    al = np.tril(a) + np.tril(a, -1).T
    res = np.abs(al @ v.T - v.T * w[None, :]).max() / scale
    orth = np.abs(v @ v.T - np.eye(n)).max()
    return np.abs(w - wr).max() / scale, res, orth


def main():
    print(ctx.info(), flush=True)
    rs = np.random.RandomState(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--wgs":
        n = int(sys.argv[2])
        a = rs.randn(n, n)
        a = a + a.T
        run(a, 1)
        for wgs in (32, 64, 128, 256):
            l0 = counter("resident_launches")
            (w, v), t = run(a, 1, wgs=wgs, reps=5)
            print(f"n={n} workgroups>={wgs}: {t * 1e3:.2f} ms, launches {counter('resident_launches') - l0}, "
                  f"errors {errors(a, w, v)}", flush=True)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--stamps":
        # (library built with -DRES_STAMPS: tools/build_res_stamps_lib.sh, SPRINGCRAFT_HIP_LIB)
        out = (C.c_ulonglong * 8)()
        for n in [int(x) for x in sys.argv[2:]]:
            a = rs.randn(n, n)
            a = a + a.T
            run(a, 1, reps=2)
            L.sc_dbg_resident_stamps(out)
            (w, v), t = run(a, 1, reps=1)
            rc = L.sc_dbg_resident_stamps(out)
            steps = max(out[7], 1)
            names = ["publish+poll", "w~.v, w, a", "norm, reflector, stores", "pass", "-", "y sums"]
            print(f"n={n}: rc {rc}, {t * 1e3:.2f} ms, steps {out[7]}, cycles (100 MHz) per step: "
                  + ", ".join(f"{names[i]} {out[i] / steps:.1f}" for i in (0, 1, 2, 3, 5))
                  + f" | total {sum(out[i] for i in range(6)) / steps:.1f} = {sum(out[i] for i in range(6)) / 100e3:.2f} ms", flush=True)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--phases":
        t6 = (C.c_double * 6)()
        ms = C.c_double(0)
        for n in [int(x) for x in sys.argv[2:]]:
            a = rs.randn(n, n)
            a = a + a.T
            for mode in (0, 1):
                run(a, mode, reps=2)
                L.sc_ctx_set_profiling(ctx.handle, 1)
                (w, v), t = run(a, mode, reps=1)
                L.sc_ctx_set_profiling(ctx.handle, 0)
                L.sc_last_eigh_timings(ctx.handle, t6)
                rc = L.sc_last_eigh_phase_ms(ctx.handle, b"resident_tridiag", C.byref(ms))
                print(f"n={n} resident={mode}: profiled solve {t * 1e3:.2f} ms: tridiagonalisation {t6[0]:.2f} "
                      f"(resident kernel {ms.value if rc == 0 else 0.0:.2f}), D&C {t6[1]:.2f}, back-transformation {t6[2]:.2f} ms",
                      flush=True)
        return
    sizes = [int(s) for s in sys.argv[1:]] or [128, 300, 900, 1536, 2048]
    for n in sizes:
        a = rs.randn(n, n)
        a = a + a.T
        (w0, v0), t0 = run(a, 0)
        l0 = counter("resident_launches")
        (w1, v1), t1 = run(a, 1)
        took = counter("resident_launches") - l0
        tnp = time.perf_counter()
        np.linalg.eigh(a)
        tnp = time.perf_counter() - tnp
        e0, e1 = errors(a, w0, v0), errors(a, w1, v1)
        print(f"n={n:5d} per-column {t0 * 1e3:7.2f} ms  resident {t1 * 1e3:7.2f} ms ({took} launches)  numpy {tnp * 1e3:7.1f} ms | "
              f"|dw| {e0[0]:.1e} / {e1[0]:.1e}  resid {e0[1]:.1e} / {e1[1]:.1e}  orth {e0[2]:.1e} / {e1[2]:.1e}  "
              f"takeovers {counter('resident_takeovers')}", flush=True)
    # the take-over routes at a small order: a failed roll call, a wait lost in mid-run
    n = 300
    a = rs.randn(n, n)
    a = a + a.T
    for hook, what in ((1, "roll call fails"), (2 + 100, "exchange of step 100 fails")):
        t_before = counter("resident_takeovers")
        (w, v), t = run(a, 1, hook=hook, reps=1)
        print(f"n={n} {what}: {t * 1e3:.1f} ms, takeovers +{counter('resident_takeovers') - t_before}, errors {errors(a, w, v)}",
              flush=True)
    (w, v), t = run(a, 1, reps=1)
    print(f"n={n} afterwards (mode 1 re-arms): {t * 1e3:.1f} ms, errors {errors(a, w, v)}", flush=True)


if __name__ == "__main__":
    main()

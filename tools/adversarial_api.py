"""Edge cases of the model / assembly API against the oracle (tiny N, coincident atoms, isolated atoms, float32 and
non-contiguous input, boundary orders of the eigensolver).  python tools/adversarial_api.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
import springcraft_amd as sc  # noqa: E402
from oracle import enm_oracle as orc  # noqa: E402
from springcraft_amd import _hip  # noqa: E402

fails = 0


def report(name, ok, extra=""):
    global fails
    fails += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} {name} {extra}", flush=True)


rs = np.random.RandomState(0)
# ---- tiny structures
for n in (1, 2, 3, 4, 7):
    coord = rs.rand(n, 3) * 5
    try:
        k, p = sc.compute_kirchhoff(coord, sc.InvariantForceField(6.0))
        kr, pr = orc.compute_kirchhoff(coord, orc.invariant_ff(6.0))
        h, _ = sc.compute_hessian(coord, sc.InvariantForceField(6.0))
        hr, _ = orc.compute_hessian(coord, orc.invariant_ff(6.0))
        ok = np.array_equal(k, kr) and np.array_equal(p.reshape(-1, 2), pr.reshape(-1, 2)) and np.allclose(h, hr, atol=1e-13)
        w = sc.ANM(coord, sc.InvariantForceField(6.0)).eigen()[0]
        wr = np.linalg.eigvalsh(hr)
        ok = ok and np.allclose(w, wr, atol=1e-12 * max(1.0, np.abs(wr).max()))
        report(f"N={n} assembly + ANM.eigen", ok)
    except Exception as e:
        report(f"N={n} assembly + ANM.eigen", False, f"{type(e).__name__}: {e}")
# ---- no contacts at all / everything isolated
coord = np.arange(30, dtype=float).reshape(10, 3) * 100.0
h, p = sc.compute_hessian(coord, sc.InvariantForceField(5.0))
report("no contacts: zero Hessian, empty pairs", np.count_nonzero(h) == 0 and len(p) == 0)
w, v = sc.ANM(coord, sc.InvariantForceField(5.0)).eigen()
report("no contacts: eigen of the zero matrix", np.all(w == 0) and np.allclose(v @ v.T, np.eye(30)))
# ---- coincident atoms: d = 0
coord = rs.rand(12, 3) * 6
coord[5] = coord[2]
for name, ff, off in (("invariant", sc.InvariantForceField(7.0), orc.invariant_ff(7.0)), ("hinsen", sc.HinsenForceField(), orc.hinsen_ff(None))):
    h, p = sc.compute_hessian(coord, ff)
    with np.errstate(all="ignore"):
        hr, pr = orc.compute_hessian(coord, off)
    same_nan = np.array_equal(np.isnan(h), np.isnan(hr))
    report(f"coincident atoms ({name}): pairs equal, NaN pattern as the reference", np.array_equal(p, pr) and same_nan and
           np.allclose(np.nan_to_num(h), np.nan_to_num(hr), atol=1e-10 * np.nanmax(np.abs(hr))))
# ---- float32 and non-contiguous coordinates
base = (rs.rand(40, 3) * 12).astype(np.float32)
h32, _ = sc.compute_hessian(base, sc.InvariantForceField(8.0))
hr32, _ = orc.compute_hessian(base.astype(np.float64), orc.invariant_ff(8.0))
report("float32 coordinates are up-cast (interaction.py:88)", np.allclose(h32, hr32, atol=1e-12))
big = rs.rand(80, 6) * 12
nc = big[::2, ::2]
hn, _ = sc.compute_hessian(nc, sc.InvariantForceField(8.0))
hrn, _ = orc.compute_hessian(np.ascontiguousarray(nc), orc.invariant_ff(8.0))
report("non-contiguous coordinates", np.allclose(hn, hrn, atol=1e-12))
# ---- wrong shapes -> ValueError as the reference (interaction.py:141-147)
for bad in (np.zeros((5, 2)), np.zeros((5,)), np.zeros((2, 5, 3))):
    try:
        sc.compute_hessian(bad, sc.InvariantForceField(8.0))
        report(f"shape {bad.shape} raises ValueError", False)
    except ValueError:
        report(f"shape {bad.shape} raises ValueError", True)
    except Exception as e:
        report(f"shape {bad.shape} raises ValueError", False, type(e).__name__)
# ---- eigensolver orders around its internal boundaries, both paths where allowed
ctx = _hip.context()
for n in (1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513):
    a = rs.standard_normal((n, n)); a = a + a.T
    wr = np.linalg.eigvalsh(a)
    for two in (False, True):
        ctx.set_two_stage(two)
        try:
            w, v = sc.nma.eigh(a)
            ok = np.abs(w - wr).max() <= 1e-12 * max(1.0, np.abs(wr).max()) and np.abs(v @ v.T - np.eye(n)).max() <= 1e-12 \
                and np.abs(a @ v.T - v.T * w[None, :]).max() <= 1e-11 * max(1.0, np.abs(wr).max())
            report(f"eigh n={n} two_stage={two}", ok)
        except Exception as e:
            report(f"eigh n={n} two_stage={two}", False, f"{type(e).__name__}: {e}")
ctx.set_two_stage(None)
print("FAILURES:", fails)
sys.exit(1 if fails else 0)

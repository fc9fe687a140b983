"""
ORACLE tooling — generates the golden vectors under ``tests/golden/generated`` by
importing the *reference itself* (``/root/reference/src/springcraft``) in the build
container.  Run here only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--big | --big-only]

``biotite`` (third-party, absent from the image) is replaced by the few-line stand-in in
``oracle/biotite_standin`` (our own code).  Every reference call below uses
``use_cell_list=False`` so that each arithmetic step on the path is the reference's own
code + NumPy (interaction.py:160-166).  Outputs are DATA only (inputs + expected
outputs); no reference source is copied.

What is written (all ``.npz``):
  structures.npz          C-alpha coords / res_name / chain_id / res_id of the reference's
                          two test PDBs (tests/data/1l2y.pdb, 7cal.pdb), model 1
  c1_1l2y_gnm7.npz        config 1: GNM Kirchhoff, pairs, eigenpairs (InvariantForceField 7.0)
  c2_n512_inv13.npz       config 2: contact counts, pair digest, Hessian digests, all
                          eigenvalues, modes 6..15
  c3_n2000_hinsen.npz     config 3 (no cutoff) and its 13 A variant: digests + all eigenvalues
  c4_n1000_inv13.npz      config 4: 4 of the 256 structures (seeds 0..3): digests + eigenvalues
  patched_n40.npz         patch semantics (shutdown / pair_off / pair_on) on a 40-atom toy
  misc_1l2y.npz           Hinsen / ParameterFree Hessians on 1l2y, mass-weighted eigenvalues
  c7cal_inv13.npz         7cal (1776 C-alpha, n=5328) Invariant 13 A eigenvalues
  c5_n8000_inv13.npz      (--big) config 5: lowest 106 eigenvalues via scipy subset_by_index
"""

import hashlib
import os
import sys
from os.path import abspath, dirname, join

import numpy as np

HERE = dirname(abspath(__file__))
REPO = dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, join(HERE, "biotite_standin"))
sys.path.insert(1, "/root/reference/src")

import springcraft as ref  # noqa: E402  (the reference, imported read-only)

OUT = join(REPO, "tests", "golden", "generated")
REF_DATA = "/root/reference/tests/data"


def synthetic_coord(n, seed, box=None):
    # the reference's own generator: tests/test_interaction.py:80-84
    if box is None:
        box = 5.0 * n ** (1.0 / 3.0)
    np.random.seed(seed)
    return np.random.rand(n, 3) * box


def read_ca(path):
    """Fixed-column PDB reader: model 1, ATOM, name CA, element C (test_anm.py:14-18)."""
    xyz, res_name, chain, res_id = [], [], [], []
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec == "ENDMDL":
                break
            if rec != "ATOM  ":
                continue
            if line[12:16].strip() != "CA" or line[76:78].strip() != "C":
                continue
            xyz.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
            res_name.append(line[17:20])
            chain.append(line[21])
            res_id.append(int(line[22:26]))
    return (
        np.array(xyz, dtype=np.float32),
        np.array(res_name),
        np.array(chain),
        np.array(res_id),
    )


def pair_digest(pairs):
    p = np.ascontiguousarray(pairs.astype(np.int64))
    return hashlib.sha256(p.tobytes()).hexdigest()


def hessian_digest(h, n, rng):
    """Frobenius norm, all diagonal 3x3 blocks, 64 sampled off-diagonal blocks, row sums."""
    h4 = h.reshape(n, 3, n, 3)
    idx = np.arange(n)
    diag = h4[idx, :, idx, :].copy()  # (n,3,3)
    si = rng.randint(0, n, size=64)
    sj = rng.randint(0, n, size=64)
    blocks = h4[si, :, sj, :].copy()
    return dict(
        fro=np.linalg.norm(h),
        diag_blocks=diag,
        sample_i=si,
        sample_j=sj,
        sample_blocks=blocks,
        row_abs_sums=np.abs(h).sum(axis=1),
    )


def save(name, **arrays):
    path = join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def anm_case(coord, ff, n_vec=(6, 16), masses=None):
    n = len(coord)
    h, pairs = ref.compute_hessian(coord, ff, use_cell_list=False)
    k, pairs_k = ref.compute_kirchhoff(coord, ff, use_cell_list=False)
    assert (pairs == pairs_k).all()
    anm = ref.ANM(coord, ff, use_cell_list=False)
    w, v = anm.eigen()
    dig = hessian_digest(h, n, np.random.RandomState(12345))
    out = dict(
        n_pairs=np.int64(len(pairs)),
        pairs_sha256=np.array(pair_digest(pairs)),
        pairs_head=pairs[:8],
        pairs_tail=pairs[-8:],
        kirchhoff_diag=np.diag(k).copy(),
        kirchhoff_offdiag_sum=np.float64(k.sum() - np.trace(k)),
        eigenvalues=w,
        eigenvectors_sel=v[n_vec[0] : n_vec[1]].copy(),
        eigenvectors_sel_range=np.array(n_vec),
    )
    out.update({f"hess_{k_}": v_ for k_, v_ in dig.items()})
    return out


def main():
    big = "--big" in sys.argv or "--big-only" in sys.argv
    os.makedirs(OUT, exist_ok=True)
    if "--big-only" in sys.argv:
        return big_case()

    # ---- structures ---------------------------------------------------------------
    s = {}
    for name in ("1l2y", "7cal"):
        xyz, rn, ch, ri = read_ca(join(REF_DATA, f"{name}.pdb"))
        s[f"{name}_coord"] = xyz
        s[f"{name}_res_name"] = rn
        s[f"{name}_chain_id"] = ch
        s[f"{name}_res_id"] = ri
        print(name, xyz.shape)
    save("structures.npz", **s)
    ca_1l2y = s["1l2y_coord"]
    ca_7cal = s["7cal_coord"]

    # ---- C1: 1l2y GNM, Invariant 7.0 (README example) -------------------------------
    ff = ref.InvariantForceField(7.0)
    k, pairs = ref.compute_kirchhoff(ca_1l2y, ff, use_cell_list=False)
    gnm = ref.GNM(ca_1l2y, ff, use_cell_list=False)
    w, v = gnm.eigen()
    save("c1_1l2y_gnm7.npz", kirchhoff=k, pairs=pairs, eigenvalues=w, eigenvectors=v)

    # ---- C2: N=512, box 40, seed 0, Invariant 13 ------------------------------------
    coord = synthetic_coord(512, 0, 40.0)
    save("c2_n512_inv13.npz", **anm_case(coord, ref.InvariantForceField(13.0)))

    # ---- C3: N=2000, Hinsen (no cutoff) and Hinsen 13 A ------------------------------
    coord = synthetic_coord(2000, 0)
    a = anm_case(coord, ref.HinsenForceField())
    b = anm_case(coord, ref.HinsenForceField(13.0))
    merged = {f"nocut_{k_}": v_ for k_, v_ in a.items()}
    merged.update({f"cut13_{k_}": v_ for k_, v_ in b.items()})
    save("c3_n2000_hinsen.npz", **merged)

    # ---- C4: N=1000, box 50, seeds 0..3, Invariant 13 -------------------------------
    merged = {}
    for seed in range(4):
        c = anm_case(synthetic_coord(1000, seed, 50.0), ref.InvariantForceField(13.0))
        merged.update({f"s{seed}_{k_}": v_ for k_, v_ in c.items()})
    save("c4_n1000_inv13.npz", **merged)

    # ---- patch semantics on a 40-atom toy ---------------------------------------------
    coord = synthetic_coord(40, 7, 14.0)
    shutdown = np.array([3, 17, 29])
    pair_off = np.array([[0, 1], [5, 9], [10, 30], [3, 4]])
    pair_on = np.array([[0, 39], [3, 20], [12, 13], [5, 9]])
    fcs = np.array([2.5, 0.75, 10.0, 3.0])
    out = dict(coord=coord, shutdown=shutdown, pair_off=pair_off, pair_on=pair_on, force_constants=fcs)
    for tag, base in (("inv6", ref.InvariantForceField(6.0)), ("hinsen8", ref.HinsenForceField(8.0)),
                      ("hinsen_nocut", ref.HinsenForceField())):
        for ptag, kw in (
            ("shutdown", dict(contact_shutdown=shutdown)),
            ("off", dict(contact_pair_off=pair_off)),
            ("on", dict(contact_pair_on=pair_on, force_constants=fcs)),
            ("all", dict(contact_shutdown=shutdown, contact_pair_off=pair_off,
                         contact_pair_on=pair_on, force_constants=fcs)),
        ):
            pff = ref.PatchedForceField(base, **kw)
            k, pairs = ref.compute_kirchhoff(coord, pff, use_cell_list=False)
            h, pairs_h = ref.compute_hessian(coord, pff, use_cell_list=False)
            out[f"{tag}_{ptag}_kirchhoff"] = k
            out[f"{tag}_{ptag}_hessian"] = h
            out[f"{tag}_{ptag}_pairs"] = pairs
    save("patched_n40.npz", **out)

    # ---- 1l2y misc: Hinsen / pfENM Hessians, mass weighting ---------------------------
    masses = np.genfromtxt(join(REF_DATA, "bio3d_mass_1l2y.csv.gz"), delimiter=",")
    out = {}
    for tag, ff in (("hinsen", ref.HinsenForceField()), ("pf", ref.ParameterFreeForceField()),
                    ("inv13", ref.InvariantForceField(13.0))):
        h, pairs = ref.compute_hessian(ca_1l2y, ff, use_cell_list=False)
        out[f"{tag}_hessian"] = h
        out[f"{tag}_pairs"] = pairs
        w, v = np.linalg.eigh(h)
        out[f"{tag}_eigenvalues"] = w
        # mass weighting exactly as anm.py:89-94,112-113
        mw = 1 / np.sqrt(masses)
        mw = np.repeat(mw, 3)
        hm = h * np.outer(mw, mw)
        out[f"{tag}_mw_hessian"] = hm
        out[f"{tag}_mw_eigenvalues"] = np.linalg.eigh(hm)[0]
    out["masses"] = masses
    save("misc_1l2y.npz", **out)

    # ---- 7cal Invariant 13 (n = 5328) -------------------------------------------------
    anm = ref.ANM(ca_7cal, ref.InvariantForceField(13.0), use_cell_list=False)
    w, v = anm.eigen()
    _, pairs = ref.compute_hessian(ca_7cal, ref.InvariantForceField(13.0), use_cell_list=False)
    save("c7cal_inv13.npz", eigenvalues=w, n_pairs=np.int64(len(pairs)),
         pairs_sha256=np.array(pair_digest(pairs)))

    if big:
        big_case()


def big_case():
    # ---- C5: N=8000, box 100, Invariant 13, lowest 106 modes --------------------------
    import scipy.linalg

    coord = synthetic_coord(8000, 0, 100.0)
    h, pairs = ref.compute_hessian(coord, ref.InvariantForceField(13.0), use_cell_list=False)
    dig = hessian_digest(h, 8000, np.random.RandomState(12345))
    w = scipy.linalg.eigh(h, eigvals_only=True, subset_by_index=[0, 105], overwrite_a=True)
    save("c5_n8000_inv13.npz", eigenvalues_low106=w, n_pairs=np.int64(len(pairs)),
         pairs_sha256=np.array(pair_digest(pairs)), pairs_head=pairs[:8], pairs_tail=pairs[-8:],
         hess_fro=dig["fro"], hess_diag_blocks=dig["diag_blocks"])


if __name__ == "__main__":
    main()

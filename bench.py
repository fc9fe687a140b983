#!/usr/bin/env python3
"""
bench.py — headline benchmark of BASELINE.json: ANM modes/s (Hessian build + full eigensolve),
N = 2000 C-alpha, HinsenForceField without cutoff (config C3), float64, all 6000 modes.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5] [--structures-per-gpu B] [--n-atoms N]

With ``--gpus N`` (N > 1) and no RANK in the environment the script starts the N ranks itself (a child
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...``, started
before this process makes any GPU call) and exits with the child's code; under torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  WORLD_SIZE != --gpus is an error.  With more than one
rank every rank pins its host threads to the cores of its GPU's NUMA node before its first GPU call.

--config c3 (default, the metric's configuration), c2 (N = 512, InvariantForceField 13 A), c5 (N = 8000, modes 0..105)
    One "step" = one pass of the hot path over one batch of B synthetic structures per GPU (c2 / c3: 64, c5: 1) whose
    coordinates are already resident in HBM: batched Hessian assembly (HIP) -> batched eigensolve (HIP), eigenvalues and
    eigenvectors left resident in HBM.  Structures are independent, so ranks share no data-path collective (weak
    scaling: B structures per GPU); RCCL is only used for the barrier and the max-over-ranks of the elapsed time.
--config c4 (BASELINE.json configs[3])
    32 * N independent N = 1000 C-alpha ANM solves (InvariantForceField 13 A) through
    ``springcraft_amd.batch.solve_sharded``: the root scatters the coordinate shards over RCCL, every rank solves its
    32 structures, the root gathers the eigenvalues.  One step = one such call (the exchange steps are inside the
    timed region); eigenvalues of every structure are checked against the oracle-independent residual on the ranks.

Rank 0 prints ONE JSON line with the driver's contract fields plus
  roofline      the kernel (group) with the longest duration in one extra profiled step right after the timed region
                (C3: k_bt2_apply, f64-MFMA bound: algorithmic flops of the launch / its duration vs the 78.6 TFLOP/s
                f64 matrix peak; one-stage path: k_symv_tiles, HBM bound).  Durations from HIP events on the solver's
                stream.  ``traffic`` only from a PMC pass collected at the benchmarked (n, batch), else null.
  rooflines     every kernel group with a model: ``assembly`` (k_hessian: 72 B per ordered atom pair / kernel time vs
                8 TB/s HBM), ``band_reduction`` / ``syr2k`` / ``symm`` / ``bt2_apply`` / ``bt1_*`` (f64-MFMA fraction),
                ``bulge_chasing`` (bytes the implementation moves AND the compulsory bytes / wall time vs 8 TB/s)
  counters      persistent-chase event counters of the solver's context (sc_ctx_get_counter): time-outs must be 0
  parity_gates  SURVEY.md section 8(d): contact counts, pair list, Kirchhoff (bit exact), Hessian (rel. Frobenius),
                eigenvalues vs the CPU run, residual and orthogonality of several structures of the timed batch
                (c5: reference-generated eigenvalues, residual of all 106 vectors, analytic rigid-body null space)
  cpu_baseline  the oracle (NumPy restatement of compute_hessian + numpy.linalg.eigh = LAPACK dsyevd, the
                reference's own driver) on this box's host cores, median over a bounded sample of ONE-structure runs
                (c5: scipy.linalg.eigh(subset_by_index) on a smaller structure, scaled by N^3 and said so)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6     # MI355X datasheet FP64 matrix (256 CUs x 4 SIMDs x 2.4 GHz x 32 flop/clk)


def symv_algorithmic_bytes(n):
    """Lower-triangle elements of the trailing matrix, read once per Householder column (8 B each)."""
    m = np.arange(n - 1, 1, -1, dtype=np.float64)   # m_c = n - c - 1 for c = 0 .. n-3
    return float(np.sum(m * (m + 1) / 2) * 8.0)


def bt2_flops(n, ncols):
    """
    k_bt2_apply, per matrix: (algorithmic, executed) flops.  Sweep s (0 .. n-3) of the bulge chase leaves reflectors
    of length min(64, n - r0) at rows r0 = s + 1 + 64 k; applying one of length L to a column costs 4 L flops.
    The kernel applies the 64 sweeps at one chase position (a "diamond", a 127 x 64 parallelogram) as four compact-WY
    blocks of 16 sweeps, each meeting 5 row tiles in both products: 4 x (20 + 20) operand fragments (16 x 4) =
    2 * 64 * 160 MFMA flops per diamond and column (round 2: one 64-sweep block, 80 + 104 fragments).
    """
    s = np.arange(0, n - 2, dtype=np.int64)
    total_len = 0
    for k in range((n - 1 + 63) // 64):
        r0 = s + 1 + 64 * k
        total_len += int(np.clip(n - r0, 0, 64).sum())
    ngroups = (n - 2 + 63) // 64
    ndia = sum((n - 1 - 64 * g + 63) // 64 for g in range(ngroups))
    return 4.0 * total_len * ncols, 2.0 * 64 * 160 * ndia * ncols


def bulge_bytes(n):
    """
    Algorithmic bytes of the bulge chase per matrix (k_bulge_step): task (sweep s, position k) owns the rows r0 .. r0+L-1,
    r0 = s + 1 + 64 k, L = min(64, n - r0); it reads and writes the L x 64 off-diagonal block (k > 0) and the lower
    triangle of the L x L diagonal block.
    """
    s = np.arange(0, n - 2, dtype=np.int64)
    total = 0
    for k in range((n - 1 + 63) // 64):
        r0 = s + 1 + 64 * k
        L = np.clip(n - r0, 0, 64)
        total += int((2 * 8 * (L * (L + 1) // 2 + (64 * L if k > 0 else 0))).sum())
    return total


def pmc_traffic(kind, n, batch):
    """
    HBM / fabric bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc
    FETCH_SIZE and WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes).  Returned only when the
    passes were collected at exactly this matrix order AND batch; None otherwise (no extrapolation).
    """
    names = {"bt2": ["r06_bt2_pmc_fetch_write.json", "r05_bt2_pmc_fetch_write.json", "r03_bt2_pmc_fetch_write.json"], "symv": ["r02_symv_pmc_fetch_size.json", "r01_symv_pmc_fetch_size.json"]}
    for name in names[kind]:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if int(d["n"]) == int(n) and int(d.get("batch", -1)) == int(batch):
                return round(float(d["hbm_bytes_per_launch_corrected"]))
        except Exception:
            continue
    return None


def synthetic_coords(n_atoms, seeds):
    # the reference's own generator (tests/test_interaction.py:80-84), density 0.008 A^-3
    box = 5.0 * n_atoms ** (1.0 / 3.0)
    out = np.empty((len(seeds), n_atoms, 3))
    for k, s in enumerate(seeds):
        out[k] = np.random.RandomState(s).rand(n_atoms, 3) * box
    return out


def host_threads():
    try:
        from threadpoolctl import threadpool_info

        return int(max([p.get("num_threads", 1) for p in threadpool_info()] + [1]))
    except Exception:
        return int(os.cpu_count() or 1)


# the CPU set this process was started with, recorded before bind_rank_to_numa_node narrows it
_START_AFFINITY = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None


class host_cores:
    """
    Context of the CPU legs (cpu_baseline*, parity gates): SURVEY.md section 8(d) asks for the oracle "on the same box's
    host cores", whatever launched this rank.  Two things stand in the way under ``torch.distributed.run``: the launcher
    exports OMP_NUM_THREADS=1 when it is unset and there is more than one rank (OpenBLAS then runs numpy.linalg.eigh on
    ONE thread), and bind_rank_to_numa_node has shrunk the affinity mask to one NUMA node for the GPU-driving phase.
    Inside the context the mask the process started with is back and the BLAS / OpenMP pools are sized to it
    (threadpoolctl sets them at run time, the environment variable is only their start-up default); both are
    restored on exit.
    """

    def __enter__(self):
        self.saved_affinity = None
        self.limits = None
        if _START_AFFINITY is not None:
            try:
                self.saved_affinity = os.sched_getaffinity(0)
                os.sched_setaffinity(0, _START_AFFINITY)
            except OSError:
                self.saved_affinity = None
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else int(os.cpu_count() or 1)
        try:
            from threadpoolctl import threadpool_limits

            self.limits = threadpool_limits(limits=ncores)
        except Exception:   # noqa: BLE001  (no threadpoolctl: the pools keep their start-up size, `cores` reports it)
            self.limits = None
        return self

    def __exit__(self, *exc):
        if self.limits is not None:
            self.limits.restore_original_limits()
        if self.saved_affinity is not None:
            try:
                os.sched_setaffinity(0, self.saved_affinity)
            except OSError:
                pass
        return False


def cpu_baseline(n_atoms, ff_name, runs=3, budget_s=12.0):
    """
    Oracle on the host cores: ONE structure of the same workload per run, at least `runs` runs and as many more as fit
    into `budget_s` seconds (bounded sample); the reported figure is the median run.
    """
    from oracle import enm_oracle as orc

    coord = synthetic_coords(n_atoms, [0])[0]
    ff = orc.hinsen_ff() if ff_name == "hinsen" else orc.invariant_ff(13.0)
    ta, te = [], []
    h = pairs = w = None
    t_begin = time.perf_counter()
    while len(ta) < runs or (time.perf_counter() - t_begin < budget_s and len(ta) < 64):
        t0 = time.perf_counter()
        h, pairs = orc.compute_hessian(coord, ff)
        t1 = time.perf_counter()
        w, _ = orc.eigen(h.copy())
        t2 = time.perf_counter()
        ta.append(t1 - t0)
        te.append(t2 - t1)
    tot = sorted(a + e for a, e in zip(ta, te))
    med = tot[len(tot) // 2]
    shown = ", ".join(f"{t:.2f}" for t in tot[:8]) + (" ..." if len(tot) > 8 else "")
    return {
        "value": 3 * n_atoms / med,
        "unit": "modes/s",
        "cores": host_threads(),
        "kind": "port",
        "sample": (f"1 structure N={n_atoms}, median of {len(tot)} runs: assembly {sorted(ta)[len(ta) // 2]:.2f} s + "
                   f"numpy.linalg.eigh {sorted(te)[len(te) // 2]:.2f} s (runs: {shown} s)"),
    }, h, pairs, w


def cpu_baseline_partial(n_atoms, n_modes, sample_atoms=3000):
    """
    Config C5 on the host: the reference itself has no partial-spectrum mode (nma.py:61 always solves for all n), so
    the baseline is the oracle's Hessian + scipy.linalg.eigh(subset_by_index) (LAPACK dsyevr), what a user of the
    reference would write.  One N = 8000 solve takes minutes on the host; the bounded sample is ONE solve at
    N = sample_atoms (same density, same force field), scaled by (N / sample_atoms)^3 -- tridiagonalisation dominates
    and is cubic -- and said so in `sample`.
    """
    from oracle import enm_oracle as orc

    coord = synthetic_coords(sample_atoms, [0])[0]
    t0 = time.perf_counter()
    h, _ = orc.compute_hessian(coord, orc.invariant_ff(13.0))
    t1 = time.perf_counter()
    try:
        import scipy.linalg

        scipy.linalg.eigh(h, subset_by_index=[0, n_modes - 1], overwrite_a=True, check_finite=False)
        how = "scipy.linalg.eigh(subset_by_index)"
    except ImportError:
        np.linalg.eigh(h)
        how = "numpy.linalg.eigh (all modes; scipy missing)"
    t2 = time.perf_counter()
    scale = (n_atoms / sample_atoms) ** 3
    est = (t2 - t0) * scale
    return {
        "value": n_modes / est, "unit": "modes/s", "cores": host_threads(), "kind": "port",
        "sample": (f"ONE solve at N={sample_atoms} (assembly {t1 - t0:.2f} s + {how} {t2 - t1:.2f} s), scaled by "
                   f"(N/{sample_atoms})^3 = {scale:.1f} to N={n_atoms}: estimated {est:.0f} s per solve"),
    }


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """
    --gpus N without a rank environment: start the N ranks as a child job (nothing here has touched the GPU).  Every rank
    binds itself to the host cores next to its GPU before its first GPU call (bind_rank_to_numa_node).
    """
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=spawn_env(args.gpus))


def spawn_env(nproc):
    """
    Environment of the rank processes.  torch.distributed.run sets OMP_NUM_THREADS=1 for every rank when the variable is
    unset and nproc > 1; give it explicitly instead: the cores this process may use, shared out over the ranks (the CPU
    baseline leg of rank 0 widens its pools to all of them at run time, see host_cores).
    """
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else int(os.cpu_count() or 1)
    env.setdefault("OMP_NUM_THREADS", str(max(1, ncores // max(1, nproc))))
    return env


def bind_rank_to_numa_node(local_rank):
    """
    One process per GPU: keep the rank's host threads (the launch thread issues ~50 k kernel launches per step) on the
    cores of the NUMA node its GPU hangs off, read from sysfs (no GPU call).  Best effort: anything missing -> no binding.
    Returns a short description for the bench line.
    """
    try:
        import glob

        # the render nodes' PCI devices in bus order = HIP's default device order on a single-node box
        devs = []
        for d in sorted(glob.glob("/sys/class/drm/renderD*/device")):
            try:
                with open(os.path.join(d, "vendor")) as f:
                    if f.read().strip() != "0x1002":
                        continue
                devs.append(os.path.realpath(d))
            except OSError:
                continue
        devs.sort()
        if local_rank >= len(devs):
            return "unbound (GPU not found in sysfs)"
        with open(os.path.join(devs[local_rank], "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return "unbound (no NUMA information)"
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return f"unbound (NUMA node {node} outside the allowed CPU set)"
        os.sched_setaffinity(0, allowed)
        return f"NUMA node {node}, {len(allowed)} cpus"
    except Exception as e:   # noqa: BLE001
        return f"unbound ({type(e).__name__})"


def parity_gates(sc, solver, coord, w_all, v_all, n_atoms, ff_name, cpu_h, cpu_pairs, cpu_w, torch):
    """SURVEY.md section 8(d) gates on the batch that was just timed (device results vs the oracle's structure 0)."""
    from oracle import enm_oracle as orc

    gates = {}
    c0 = coord[0].cpu().numpy()
    # contact scan + pair list + Kirchhoff with an integer-valued force field: bit exact
    k_gpu, p_gpu = sc.compute_kirchhoff(c0, sc.InvariantForceField(13.0))
    k_cpu, p_cpu = orc.compute_kirchhoff(c0, orc.invariant_ff(13.0))
    gates["kirchhoff_inv13_bit_exact"] = bool(np.array_equal(k_gpu, k_cpu))
    gates["contact_counts_inv13_equal"] = bool(np.array_equal(np.diag(k_gpu).astype(np.int64), np.diag(k_cpu).astype(np.int64)))
    gates["pairs_inv13_equal"] = bool(np.array_equal(p_gpu, p_cpu))
    # the benchmarked force field: pair list and Hessian of structure 0
    h_gpu, p_gpu = sc.compute_hessian(c0, sc.HinsenForceField() if ff_name == "hinsen" else sc.InvariantForceField(13.0))
    gates["pairs_equal"] = bool(np.array_equal(p_gpu, cpu_pairs))
    gates["n_pairs"] = int(len(p_gpu))
    gates["hessian_rel_frobenius"] = float(np.linalg.norm(h_gpu - cpu_h) / np.linalg.norm(cpu_h))
    # eigenvalues of structure 0 against LAPACK on the oracle's Hessian
    w0 = w_all[0].cpu().numpy()
    lam_max = float(np.abs(cpu_w).max())
    gates["eigenvalues_max_rel_diff_nontrivial"] = float((np.abs(w0[6:] - cpu_w[6:]) / np.abs(cpu_w[6:])).max())
    gates["trivial_modes_max_abs_over_lambda_max"] = float(np.abs(w0[:6]).max() / lam_max)
    # residual and orthogonality of several structures of the batch (device arithmetic on the solver's outputs)
    B = w_all.shape[0]
    idx = sorted(set([0, B // 3, (2 * B) // 3, B - 1]))
    w_keep = w_all[idx].clone()
    v_keep = v_all[idx].clone()
    h_all = solver.assemble(coord)           # eigh destroyed the matrices: assemble again
    res, orth = [], []
    eye = torch.eye(w_all.shape[1], dtype=torch.float64, device=w_all.device)
    for j, b in enumerate(idx):
        h, w, v = h_all[b], w_keep[j], v_keep[j]
        r = h @ v.T - v.T * w[None, :]
        res.append(float(torch.linalg.vector_norm(r, dim=0).max() / w.abs().max()))
        orth.append(float((v @ v.T - eye).abs().max()))
    gates["structures_checked"] = idx
    gates["residual_max_over_norm"] = max(res)
    gates["orthogonality_max"] = max(orth)
    gates["pass"] = bool(
        gates["kirchhoff_inv13_bit_exact"] and gates["pairs_inv13_equal"] and gates["pairs_equal"]
        and gates["hessian_rel_frobenius"] <= 1e-12 and gates["eigenvalues_max_rel_diff_nontrivial"] <= 1e-5
        and gates["trivial_modes_max_abs_over_lambda_max"] <= 1e-9 and gates["residual_max_over_norm"] <= 1e-5
        and gates["orthogonality_max"] <= 1e-8)
    return gates


def parity_gates_partial(solver, coord, w, v, torch):
    """
    Config C5 gates: eigenvalues against the reference-generated vector (tests/golden/generated/c5_n8000_inv13.npz, made
    by importing the reference in the build container), residual of every returned vector with the Hessian applied on
    the device, orthonormality, and the six trivial modes against the analytic rigid-body null space.
    """
    gates = {}
    n_atoms = coord.shape[1]
    n = 3 * n_atoms
    wd, vd = w[0], v[0]
    w_np, v_np = wd.cpu().numpy(), vd.cpu().numpy()
    ref_file = os.path.join(ROOT, "tests", "golden", "generated", "c5_n8000_inv13.npz")
    if n_atoms == 8000 and os.path.exists(ref_file):
        ref = np.load(ref_file)["eigenvalues_low106"][: len(w_np)]
        gates["eigenvalues_max_rel_diff_nontrivial"] = float((np.abs(w_np[6:] - ref[6:]) / np.abs(ref[6:])).max())
    h = solver.assemble(coord)[0]
    x = torch.ones(n, dtype=torch.float64, device=h.device)
    for _ in range(30):           # power iteration: ||H||_2 from below
        x = h @ x
        x = x / torch.linalg.vector_norm(x)
    lam_max = float(x @ (h @ x))
    r = h @ vd.T - vd.T * wd[None, :]
    gates["residual_max_over_norm"] = float(torch.linalg.vector_norm(r, dim=0).max() / lam_max)
    gates["orthogonality_max"] = float((vd @ vd.T - torch.eye(len(w_np), dtype=torch.float64, device=h.device)).abs().max())
    gates["trivial_modes_max_abs_over_lambda_max"] = float(np.abs(w_np[:6]).max() / lam_max)
    c = coord[0].cpu().numpy()
    basis = np.zeros((n, 6))
    for a in range(3):
        basis[a::3, a] = 1.0
        e = np.zeros(3)
        e[a] = 1.0
        basis[:, 3 + a] = np.cross(e[None, :], c - c.mean(0)).reshape(-1)
    q, _ = np.linalg.qr(basis)
    sv = np.linalg.svd(q.T @ v_np[:6].T, compute_uv=False)
    gates["null_space_singular_values_max_dev"] = float(np.abs(sv - 1.0).max())
    gates["other_modes_max_overlap_with_null_space"] = float(np.abs(q.T @ v_np[6:].T).max())
    gates["pass"] = bool(gates.get("eigenvalues_max_rel_diff_nontrivial", 0.0) <= 1e-5
                         and gates["residual_max_over_norm"] <= 1e-5 and gates["orthogonality_max"] <= 1e-8
                         and gates["trivial_modes_max_abs_over_lambda_max"] <= 1e-9
                         and gates["null_space_singular_values_max_dev"] <= 1e-6
                         and gates["other_modes_max_overlap_with_null_space"] <= 1e-6)
    return gates


def bulge_compulsory_bytes(n):
    """What band -> tridiagonal must move per matrix whatever the schedule: the band in (65 x n) and the reflectors out."""
    s = np.arange(0, n - 2, dtype=np.int64)
    refl = 0
    for k in range((n - 1 + 63) // 64):
        refl += int(np.clip(n - (s + 1 + 64 * k), 0, 64).sum())
    return 8 * (65 * n + refl)


CHASE_COUNTERS = ("chase_launches", "chase_pair_launches", "chase_timeouts", "chase_resumed", "stepwise_chases",
                  "gemm3_launches", "symm3_launches", "panel_coop_launches", "panel_coop_timeouts")


def chase_form(before, after):
    """Which form of the bulge chase a solve took, from the context's counters before / after it."""
    if after["chase_pair_launches"] > before["chase_pair_launches"]:
        return "pair"
    if after["chase_launches"] > before["chase_launches"]:
        return "sweep_per_workgroup"
    return "per_wavefront"


def build_rooflines(t, n, B, ncols, form="per_wavefront"):
    """
    (dominant-kernel roofline, all rooflines) from the phase durations `t` of one profiled step.  n: matrix order,
    B: matrices per launch, ncols: eigenvector columns the back-transformations are applied to.
    Algorithmic work per matrix: SYMM X = A22 V and the trailing SYR2K 2/3 n^3 flops each; k_bt2_apply 4 L
    flops per reflector of length L and column; stage-1 back-transformation n^2 ncols per GEMM (W = V^T Z; Z -= (V T) W);
    bulge chasing see bulge_bytes / bulge_compulsory_bytes; one-stage SYMV 8 B per lower-triangle element and column.
    """
    cand = {}

    def mfma(name, kernel, flops, ms, what, extra=None):
        if not ms or ms <= 0:
            return
        a = flops / (ms * 1e-3) / 1e12
        cand[name] = {"kernel": kernel, "bound": "mfma", "achieved": round(a, 2), "peak": F64_MFMA_PEAK_TF,
                      "unit": "TFLOP/s", "frac": round(a / F64_MFMA_PEAK_TF, 4), "ms": round(ms, 3), "what": what}
        if extra:
            cand[name].update(extra)

    n3 = float(n) ** 3
    if t.get("two_stage"):
        alg, executed = bt2_flops(n, ncols)
        mfma("bt2_apply", "k_bt2_apply", alg * B, t.get("bt2_apply_ms"),
             "stage-2 back-transformation: 4 L flops per reflector and eigenvector column / duration of the ONE launch",
             {"launches_per_step": 1, "algorithmic_flops_per_launch": alg * B, "executed_flops_per_launch": executed * B,
              "executed_over_algorithmic": round(executed / alg, 4) if alg else None,
              "executed_tflops": round(executed * B / (t["bt2_apply_ms"] * 1e-3) / 1e12, 2) if t.get("bt2_apply_ms") else None,
              "traffic": pmc_traffic("bt2", n, B) if ncols == n else None})
        mfma("band_reduction", "stage 1 (panel QR + SYMM + Gram + W + SYR2K)", 4.0 / 3.0 * n3 * B, t.get("band_reduction_ms"),
             "4/3 n^3 flops per matrix / duration of the whole stage incl. the panel QRs")
        mfma("syr2k", "k_gemm2 / k_gemm3 (trailing SYR2K launches)", 2.0 / 3.0 * n3 * B, t.get("syr2k_ms"),
             "trailing SYR2K launches of the band reduction alone: 2/3 n^3 flops per matrix / their duration")
        mfma("symm", "k_symm3 (X = A22 V, one launch per panel; the last panels and odd orders: k_gemm2 triangular-operand launches)",
             2.0 / 3.0 * n3 * B, t.get("symm_ms"),
             "X = A22 V on the lower-stored A22: 2/3 n^3 flops per matrix / duration of those launches (incl. the pair corrections of X)")
        mfma("bt1_w", "k_gemm3 (W = V^T Z)", float(n) * n * ncols * B, t.get("bt1_w_ms"),
             "stage-1 back-transformation, first product: n^2 ncols flops per matrix")
        mfma("bt1_update", "k_gemm3 (Z -= (V T) W)", float(n) * n * ncols * B, t.get("bt1_update_ms"),
             "stage-1 back-transformation, second product: n^2 ncols flops per matrix")
        if t.get("dc_gemm_gflop", 0) > 0:
            mfma("dc_gemm", "k_gemm2 (D&C merge products, gathered operands)", t["dc_gemm_gflop"] * 1e9, t.get("dc_gemm_ms"),
                 "eigenvector updates of the divide & conquer merges: 2 m n k per merge record, summed on the device (K and N "
                 "are what the deflation leaves); dense upper bound 4/3 n^3 per matrix",
                 {"dense_bound_flops": 4.0 / 3.0 * n3 * B, "flops_over_dense_bound": round(t["dc_gemm_gflop"] * 1e9 / (4.0 / 3.0 * n3 * B), 4)})
        if t.get("bulge_chasing_ms", 0) > 0:
            # pair form: one block set is read (team A) and one written (team B) per PAIR of tasks -- half the bytes
            bb = bulge_bytes(n) * B // (2 if form == "pair" else 1)
            cb = bulge_compulsory_bytes(n) * B
            ms = t["bulge_chasing_ms"]
            bw = bb / (ms * 1e-3) / 1e9
            cand["bulge_chasing"] = {
                "kernel": {"pair": "k_bulge_pair", "sweep_per_workgroup": "k_bulge_chase"}.get(form, "k_bulge_step"),
                "form": form, "bound": "hbm", "achieved": round(bw, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(bw / HBM_PEAK_GBS, 4), "ms": round(ms, 3),
                "executed_bytes_per_step": bb, "compulsory_bytes_per_step": cb,
                "compulsory_frac": round(cb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "executed_over_compulsory": round(bb / cb, 1),
                "what": "stage 2 (band -> tridiagonal).  `achieved` counts what the implementation moves (every task reads "
                        "and writes its off-diagonal block and the lower triangle of its diagonal block; in the pair form "
                        "one workgroup chases two sweeps through LDS: one read + one write per PAIR of tasks); "
                        "`compulsory_*` counts only the band in + the reflectors out, the bytes any schedule must move.  "
                        "Wall time of the stage (HIP events; parts of the batch run on separate streams, so per-kernel "
                        "durations in a rocprof summary overlap and add up to more)",
            }
    else:
        launches = max(n - 2, 1)
        if t.get("symv_ms", 0) > 0:
            bytes_per_launch = symv_algorithmic_bytes(n) * B / launches   # batched launch: B matrices
            avg_ms = t["symv_ms"] / launches
            a = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            cand["symv"] = {
                "kernel": "k_symv_tiles", "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4), "ms": round(t["symv_ms"], 3),
                "traffic": pmc_traffic("symv", n, B), "launches_per_step": launches,
                "algorithmic_bytes_per_launch": round(bytes_per_launch), "avg_launch_ms": round(avg_ms, 5),
                "what": "one-stage tridiagonalisation: y = A22 v, 8 B per lower-triangle element per column",
            }
        mfma("syr2k", "k_gemm2 (SYR2K)", 2.0 / 3.0 * n3 * B, t.get("syr2k_ms"), "panel SYR2K launches: 2/3 n^3 flops per matrix")
        if t.get("dc_gemm_gflop", 0) > 0:
            mfma("dc_gemm", "k_gemm2 (D&C merge products, gathered operands)", t["dc_gemm_gflop"] * 1e9, t.get("dc_gemm_ms"),
                 "eigenvector updates of the divide & conquer merges: 2 m n k per merge record, summed on the device")
        mfma("bt1_w", "k_gemm3 (W = V^T Z)", float(n) * n * ncols * B, t.get("bt1_w_ms"), "back-transformation, first product")
        mfma("bt1_update", "k_gemm3 (Z -= (V T) W)", float(n) * n * ncols * B, t.get("bt1_update_ms"),
             "back-transformation, second product")
    # dominant = the single kernel (group) with the longest duration; the band reduction as a whole is a stage, not a kernel
    kernels = {k: v for k, v in cand.items() if k != "band_reduction"}
    if not kernels:
        return None, cand
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    roofline = dict(kernels[dom])
    roofline["name"] = dom
    roofline.setdefault("traffic", None)
    roofline["avg_launch_ms"] = roofline.get("avg_launch_ms", roofline["ms"] / roofline.get("launches_per_step", 1))
    roofline["measured"] = "HIP events on the solver's stream around the kernel (group), one extra profiled step after the timed region"
    return roofline, cand


BATCH_CONFIGS = {
    # name: (n_atoms, force field, structures per GPU per step, subset, metric, workload text)
    "c2": (512, "inv13", 64, None, "ANM modes/sec (Hessian build + full eigensolve), N=512 C-alpha (config C2)",
           "C2: N={n_atoms} C-alpha ANM, InvariantForceField 13 A, full {n}x{n} eigensolve, all modes + vectors"),
    "c3": (2000, "hinsen", 64, None, "ANM modes/sec (Hessian build + full eigensolve), N=2000 C-alpha",
           "C3: N={n_atoms} C-alpha ANM, HinsenForceField (no cutoff), full {n}x{n} eigensolve, all modes + vectors"),
    "c5": (8000, "inv13", 1, (0, 105), "ANM modes/sec, N=8000 C-alpha, lowest 106 modes only (config C5, partial spectrum)",
           "C5: N={n_atoms} C-alpha ANM, InvariantForceField 13 A, {n}x{n} Hessian, modes 0..105 (6 trivial + 100) + vectors"),
}


def run_batch(args, rank, world, torch, dist):
    """Configs c2 / c3 / c5: B structures per GPU per step, coordinates resident in HBM, no data-path collective."""
    import springcraft_amd as sc
    from springcraft_amd.batch import DeviceBatchSolver

    n_atoms_d, ff_name, B_d, subset, metric, workload = BATCH_CONFIGS[args.config]
    n_atoms = args.n_atoms if args.n_atoms else n_atoms_d
    B = args.structures_per_gpu if args.structures_per_gpu_set else B_d
    n = 3 * n_atoms
    seeds = [rank * B + k for k in range(B)]
    box = 100.0 if (args.config == "c5" and n_atoms == 8000) else None
    coords_np = synthetic_coords(n_atoms, seeds) if box is None else np.stack(
        [np.random.RandomState(s).rand(n_atoms, 3) * box for s in seeds])
    coord = torch.from_numpy(coords_np).cuda()
    ff = sc.HinsenForceField() if ff_name == "hinsen" else sc.InvariantForceField(13.0)
    solver = DeviceBatchSolver(n_atoms, B, ff, dim=3, want_vectors=True, subset_by_index=subset)
    nmodes = n if subset is None else subset[1] - subset[0] + 1

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.solve(coord)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.solve(coord)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        # ---- one extra profiled step: durations of the kernel groups (HIP events, solver's stream)
        solver.set_profiling(True)
        c_before = {k: solver.ctx.counter(k) for k in CHASE_COUNTERS}
        w, v = solver.solve(coord)
        torch.cuda.synchronize()
        t = solver.last_timings()
        solver.set_profiling(False)
        form = chase_form(c_before, {k: solver.ctx.counter(k) for k in CHASE_COUNTERS})
        roofline, rooflines = build_rooflines(t, n, B, nmodes, form)
        # ---- assembly roofline: k_hessian alone (one launch per assemble), events on the same stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        solver.assemble(coord)
        e0.record()
        for _ in range(reps):
            solver.assemble(coord)
        e1.record()
        torch.cuda.synchronize()
        asm_ms = e0.elapsed_time(e1) / reps
        asm_bytes = (72.0 * n_atoms * n_atoms + 24.0 * n_atoms) * B
        asm = asm_bytes / (asm_ms * 1e-3) / 1e9
        rooflines["assembly"] = {
            "kernel": "k_hessian", "bound": "hbm", "achieved": round(asm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(asm / HBM_PEAK_GBS, 4), "avg_launch_ms": round(asm_ms, 4),
            "algorithmic_bytes_per_launch": round(asm_bytes),
            "what": "72 B per ordered atom pair (9 N^2 f64 written once) + 24 N B read, per structure",
        }
        counters = {k: solver.ctx.counter(k) for k in CHASE_COUNTERS}

        total_structures = B * world * args.steps
        value = nmodes * total_structures / elapsed
        cpu = gates = None
        if not args.no_cpu_baseline:
            with host_cores():
                if subset is None:
                    cpu, cpu_h, cpu_pairs, cpu_w = cpu_baseline(n_atoms, ff_name, runs=args.cpu_runs)
                    gates = parity_gates(sc, solver, coord, w, v, n_atoms, ff_name, cpu_h, cpu_pairs, cpu_w, torch)
                else:
                    gates = parity_gates_partial(solver, coord, w.clone(), v.clone(), torch)
                    cpu = cpu_baseline_partial(n_atoms, nmodes)
        out = {
            "metric": metric,
            "value": round(value, 1), "unit": "modes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": workload.format(n_atoms=n_atoms, n=n),
                "structures_per_gpu_per_step": B,
                "solves_per_s": round(total_structures / elapsed, 3),
                "parallelism": "independent structures sharded over GPUs, no data-path collective",
                "host_binding": args.host_binding,
            },
            "roofline": roofline, "rooflines": rooflines, "phases_ms_profiled_step": dict(t), "counters": counters,
            "parity_gates": gates, "cpu_baseline": cpu,
        }
    return out


def run_c4(args, rank, world, torch, dist):
    """BASELINE.json configs[3]: 32 structures of N = 1000 per GPU through solve_sharded (RCCL scatter / gather)."""
    import springcraft_amd as sc
    from springcraft_amd.batch import DeviceBatchSolver, shard_bounds, solve_sharded

    n_atoms, per_gpu = 1000, args.structures_per_gpu if args.structures_per_gpu_set else 32
    n = 3 * n_atoms
    total = per_gpu * world
    ff = sc.InvariantForceField(13.0)
    coords = synthetic_coords(n_atoms, list(range(total))) if rank == 0 else None
    lo, hi = shard_bounds(total, world, rank)
    solver = DeviceBatchSolver(n_atoms, hi - lo, ff, dim=3, want_vectors=True)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    w = None
    for _ in range(args.warmup):
        w, _ = solve_sharded(coords, ff, dim=3, want_vectors=True, solver=solver)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w, _ = solve_sharded(coords, ff, dim=3, want_vectors=True, solver=solver)
    barrier()
    elapsed = time.perf_counter() - t0
    cdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # every rank checks its own shard: residual + orthogonality of its first and last structure (device arithmetic);
    # the shard's coordinates are regenerated from the seeds (the scatter's own correctness is what w is checked for below)
    local = torch.from_numpy(synthetic_coords(n_atoms, list(range(lo, hi)))).cuda()
    w_keep, v_keep = solver.w.clone(), [solver.v[b].clone() for b in sorted(set([0, hi - lo - 1]))]
    h_all = solver.assemble(local)
    worst = torch.zeros(2, dtype=torch.float64, device="cuda")
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    for j, b in enumerate(sorted(set([0, hi - lo - 1]))):
        hb, wb, vb = h_all[b], w_keep[b], v_keep[j]
        r = hb @ vb.T - vb.T * wb[None, :]
        worst[0] = max(worst[0], torch.linalg.vector_norm(r, dim=0).max() / wb.abs().max())
        worst[1] = max(worst[1], (vb @ vb.T - eye).abs().max())
    worst = worst.to(cdev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    out = None
    if rank == 0:
        # one extra profiled local solve of rank 0's shard: kernel-group durations -> rooflines
        solver.set_profiling(True)
        c_before = {k: solver.ctx.counter(k) for k in CHASE_COUNTERS}
        solver.solve(local)
        torch.cuda.synchronize()
        tt = solver.last_timings()
        solver.set_profiling(False)
        form = chase_form(c_before, {k: solver.ctx.counter(k) for k in CHASE_COUNTERS})
        roofline, rooflines = build_rooflines(tt, n, hi - lo, n, form)
        counters = {k: solver.ctx.counter(k) for k in CHASE_COUNTERS}
        cpu = gates = None
        if not args.no_cpu_baseline:
            from oracle import enm_oracle as orc

            with host_cores():
                cpu, _, _, cpu_w = cpu_baseline(n_atoms, "inv13", runs=args.cpu_runs)
            w0 = w[0]
            gates = {
                "eigenvalues_max_rel_diff_nontrivial": float((np.abs(w0[6:] - cpu_w[6:]) / np.abs(cpu_w[6:])).max()),
                "trivial_modes_max_abs_over_lambda_max": float(np.abs(w0[:6]).max() / np.abs(cpu_w).max()),
                "residual_max_over_norm_all_ranks": float(worst[0]),
                "orthogonality_max_all_ranks": float(worst[1]),
            }
            # the gathered eigenvalues of the LAST structure come from the last rank: check them against the oracle too
            with host_cores():
                hl, _ = orc.compute_hessian(coords[-1], orc.invariant_ff(13.0))
                wl = np.linalg.eigvalsh(hl)
            gates["last_structure_eigenvalues_max_rel_diff"] = float((np.abs(w[-1][6:] - wl[6:]) / np.abs(wl[6:])).max())
            gates["pass"] = bool(gates["eigenvalues_max_rel_diff_nontrivial"] <= 1e-5
                                 and gates["last_structure_eigenvalues_max_rel_diff"] <= 1e-5
                                 and gates["residual_max_over_norm_all_ranks"] <= 1e-5
                                 and gates["orthogonality_max_all_ranks"] <= 1e-8)
        out = {
            "metric": "ANM modes/sec, batch of independent N=1000 C-alpha solves sharded over GPUs (config C4)",
            "value": round(3 * n_atoms * total * args.steps / elapsed, 1), "unit": "modes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"C4: {total} independent N={n_atoms} C-alpha ANM (InvariantForceField 13 A), all modes + vectors, "
                            f"{per_gpu} per GPU, solve_sharded: RCCL scatter of coordinates + gather of eigenvalues inside the step",
                "structures_per_gpu_per_step": per_gpu,
                "solves_per_s": round(total * args.steps / elapsed, 3),
                "parallelism": "independent structures sharded over GPUs; collectives only for scatter / gather",
                "backend": dist.get_backend(),
                "host_binding": args.host_binding,
            },
            "roofline": roofline, "rooflines": rooflines, "phases_ms_profiled_step": dict(tt), "counters": counters,
            "parity_gates": gates, "cpu_baseline": cpu,
        }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c3")
    ap.add_argument("--structures-per-gpu", type=int, default=None)
    ap.add_argument("--n-atoms", type=int, default=0, help="override the configuration's atom count (c2 / c3 / c5)")
    ap.add_argument("--cpu-runs", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    args.structures_per_gpu_set = args.structures_per_gpu is not None
    if args.structures_per_gpu is None:
        args.structures_per_gpu = 64
    args.host_binding = "not bound"

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))       # decided before anything in this process touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    # before the first GPU call of this rank: keep its host threads next to its GPU
    if os.environ.get("SPRINGCRAFT_BENCH_NO_BIND") is None and (world > 1 or os.environ.get("SPRINGCRAFT_BENCH_BIND") == "1"):
        args.host_binding = bind_rank_to_numa_node(local_rank)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: springcraft_amd has no CPU fallback")
    # one process per GPU.  Rehearsal on a box with fewer GPUs than ranks (tests / the 1-GPU development box):
    # SPRINGCRAFT_BENCH_SHARE_GPUS=1 maps rank r to device r % device_count and uses gloo (RCCL refuses two ranks on
    # one device); never set by the driver.
    share = os.environ.get("SPRINGCRAFT_BENCH_SHARE_GPUS") == "1"
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not share:
        raise SystemExit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {ndev} visible)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    if world > 1 or args.config == "c4":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("bench.py: WORLD_SIZE > 1 without MASTER_PORT (start the ranks with torch.distributed.run)")
            os.environ["MASTER_PORT"] = str(free_port())   # single-process group of --config c4: any free port will do
        if share and world > ndev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))

    out = (run_c4 if args.config == "c4" else run_batch)(args, rank, world, torch, dist)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()   # rank 0 spends tens of seconds in the CPU baseline; leave together
        dist.destroy_process_group()
    if rank == 0 and out.get("parity_gates") and not out["parity_gates"]["pass"]:
        sys.exit(3)


if __name__ == "__main__":
    main()

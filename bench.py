#!/usr/bin/env python3
"""
bench.py — headline benchmark of BASELINE.json: ANM modes/s (Hessian build + full eigensolve),
N = 2000 C-alpha, HinsenForceField without cutoff (config C3), float64, all 6000 modes.

    python bench.py --gpus N --steps K --warmup W [--structures-per-gpu B] [--n-atoms 2000]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of B synthetic structures per GPU whose
coordinates are already resident in HBM: batched Hessian assembly (HIP) -> batched eigensolve (HIP),
eigenvalues and eigenvectors left resident in HBM.  Structures are independent, so ranks share no
data-path collective (weak scaling: B structures per GPU); RCCL is only used for the barrier and the
max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line with the driver's contract fields plus
  roofline      dominant kernel.  With 64 structures per GPU the eigensolver takes its two-stage path and the
                dominant kernel is k_bt2_fused (back-transformation of the bulge-chasing reflectors, f64-MFMA
                bound): algorithmic flops (applying each reflector of length L to the 6000 eigenvector columns,
                4 L flops per column) / kernel time vs the 78.6 TFLOP/s f64 matrix peak.  On the one-stage
                path (few structures in flight) it is k_symv_tiles (HBM-bound): algorithmic bytes / kernel
                time vs 8 TB/s.  Kernel time from HIP events around the kernel's launches on the solver's
                stream, taken in one extra profiled step right after the timed region (profiling adds host
                syncs, so it is kept out of the timed steps)
  cpu_baseline  the oracle (NumPy restatement of compute_hessian + numpy.linalg.eigh = LAPACK dsyevd,
                the reference's own driver) timed on this box's host cores on ONE structure
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6     # MI355X datasheet FP64 matrix; measured issue ceiling 47 TF (profiles/r01_probe_f64.txt)


def symv_algorithmic_bytes(n):
    """Lower-triangle elements of the trailing matrix, read once per Householder column (8 B each)."""
    m = np.arange(n - 1, 1, -1, dtype=np.float64)   # m_c = n - c - 1 for c = 0 .. n-3
    return float(np.sum(m * (m + 1) / 2) * 8.0)


def bt2_flops(n, ncols):
    """
    k_bt2_fused, per matrix: (algorithmic, executed) flops.  Sweep s (0 .. n-3) of the bulge chase leaves reflectors
    of length min(64, n - r0) at rows r0 = s + 1 + 64 k; applying one of length L to a column costs 4 L flops.
    The kernel applies them 64 sweeps at a time as compact-WY "diamonds" (127 x 64 parallelograms) and skips the
    k-steps that only meet structural zeros: 2 * 64 * (80 + 64 + 40) MFMA flops per diamond and column.
    """
    s = np.arange(0, n - 2, dtype=np.int64)
    total_len = 0
    for k in range((n - 1 + 63) // 64):
        r0 = s + 1 + 64 * k
        total_len += int(np.clip(n - r0, 0, 64).sum())
    ngroups = (n - 2 + 63) // 64
    ndia = sum((n - 1 - 64 * g + 63) // 64 for g in range(ngroups))
    return 4.0 * total_len * ncols, 2.0 * 64 * (80 + 64 + 40) * ndia * ncols


def bt2_traffic_per_launch(n, batch):
    """
    HBM / fabric bytes per k_bt2_fused launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE
    and WRITE_SIZE in separate runs; FETCH_SIZE x 2 as calibrated with tools/probe_fetch_width.hip for this access
    width).  Only valid for the matrix order it was collected at; None otherwise.
    """
    path = os.path.join(ROOT, "profiles", "r01_bt2_pmc_fetch_write.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if int(d["n"]) != int(n):
            return None
        return round(float(d["hbm_bytes_per_launch_per_matrix_corrected"]) * batch)
    except Exception:
        return None


def symv_traffic_per_launch(n, batch):
    """
    HBM bytes per k_symv_tiles launch from the PMC pass committed under profiles/ (rocprofv3 --pmc
    FETCH_SIZE in its own run, x2 gfx950 correction for 16-B-per-lane reads; see that file for the command).
    Only valid for the matrix order it was collected at; None otherwise.
    """
    path = os.path.join(ROOT, "profiles", "r01_symv_pmc_fetch_size.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if int(d["n"]) != int(n):
            return None
        return round(float(d["hbm_read_bytes_per_launch_per_matrix_corrected"]) * batch)
    except Exception:
        return None


def synthetic_coords(n_atoms, seeds):
    # the reference's own generator (tests/test_interaction.py:80-84), density 0.008 A^-3
    box = 5.0 * n_atoms ** (1.0 / 3.0)
    out = np.empty((len(seeds), n_atoms, 3))
    for k, s in enumerate(seeds):
        out[k] = np.random.RandomState(s).rand(n_atoms, 3) * box
    return out


def cpu_baseline(n_atoms):
    """Oracle on the host cores: one structure of the same workload (bounded sample)."""
    from oracle import enm_oracle as orc

    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    coord = synthetic_coords(n_atoms, [0])[0]
    t0 = time.perf_counter()
    h, _ = orc.compute_hessian(coord, orc.hinsen_ff())
    t1 = time.perf_counter()
    w, v = orc.eigen(h)
    t2 = time.perf_counter()
    return {
        "value": 3 * n_atoms / (t2 - t0),
        "unit": "modes/s",
        "cores": int(cores),
        "kind": "port",
        "sample": f"1 structure N={n_atoms}: assembly {t1 - t0:.2f} s + numpy.linalg.eigh {t2 - t1:.2f} s",
    }, w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--structures-per-gpu", type=int, default=64)
    ap.add_argument("--n-atoms", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: springcraft_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import springcraft_amd as sc
    from springcraft_amd.batch import DeviceBatchSolver

    n_atoms, B = args.n_atoms, args.structures_per_gpu
    n = 3 * n_atoms
    seeds = [rank * B + k for k in range(B)]
    coord = torch.from_numpy(synthetic_coords(n_atoms, seeds)).cuda()
    solver = DeviceBatchSolver(n_atoms, B, sc.HinsenForceField(), dim=3, want_vectors=True)

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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel: one extra profiled step (rank 0 only) -------------------------
    roofline = None
    phases = None
    w_gpu0 = None
    if rank == 0:
        solver.set_profiling(True)
        w, v = solver.solve(coord)
        torch.cuda.synchronize()
        t = solver.last_timings()
        solver.set_profiling(False)
        phases = dict(t)
        if t.get("two_stage"):
            alg, executed = bt2_flops(n, n)
            ms = t["bt2_fused_ms"]
            achieved = alg * B / (ms * 1e-3) / 1e12
            roofline = {
                "kernel": "k_bt2_fused",
                "bound": "mfma",
                "achieved": round(achieved, 2),
                "peak": F64_MFMA_PEAK_TF,
                "unit": "TFLOP/s",
                "frac": round(achieved / F64_MFMA_PEAK_TF, 4),
                "traffic": bt2_traffic_per_launch(n, B),
                "launches_per_step": 1,
                "algorithmic_flops_per_launch": alg * B,
                "executed_flops_per_launch": executed * B,
                "executed_tflops": round(executed * B / (ms * 1e-3) / 1e12, 2),
                "avg_launch_ms": round(ms, 3),
                "measured": "HIP events around the launch, one extra profiled step after the timed region",
            }
            # stage 1 (band reduction): 4/3 n^3 flops per matrix in GEMMs + the panel QR
            phases["band_reduction_tflops"] = round(4.0 / 3.0 * float(n) ** 3 * B / (t["band_reduction_ms"] * 1e-3) / 1e12, 2)
        else:
            launches = (n - 2)
            bytes_per_launch = symv_algorithmic_bytes(n) * B / launches   # batched launch: B matrices
            avg_ms = t["symv_ms"] / launches
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roofline = {
                "kernel": "k_symv_tiles",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": symv_traffic_per_launch(n, B),
                "launches_per_step": launches,
                "algorithmic_bytes_per_launch": round(bytes_per_launch),
                "avg_launch_ms": round(avg_ms, 5),
                "measured": "HIP events around every launch, one extra profiled step after the timed region",
            }
            syr2k_flops = 2.0 / 3.0 * float(n) ** 3 * B
            phases["syr2k_tflops"] = round(syr2k_flops / (t["syr2k_ms"] * 1e-3) / 1e12, 2) if t["syr2k_ms"] > 0 else None
            phases["syr2k_frac_of_f64_mfma_peak"] = round(phases["syr2k_tflops"] / F64_MFMA_PEAK_TF, 4) if phases["syr2k_tflops"] else None
        w_gpu0 = w[0].cpu().numpy()

    if world > 1:
        dist.barrier()

    if rank == 0:
        total_structures = B * world * args.steps
        value = 3 * n_atoms * total_structures / elapsed
        cpu = None
        if not args.no_cpu_baseline:
            cpu, w_cpu = cpu_baseline(n_atoms)
            rel = np.abs(w_gpu0[6:] - w_cpu[6:]) / np.abs(w_cpu[6:])
            cpu["gpu_vs_cpu_max_rel_eigenvalue_diff"] = float(rel.max())
        out = {
            "metric": "ANM modes/sec (Hessian build + full eigensolve), N=2000 C-alpha",
            "value": round(value, 1),
            "unit": "modes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C3: N={n_atoms} C-alpha ANM, HinsenForceField (no cutoff), full {n}x{n} eigensolve, all modes + vectors",
                "structures_per_gpu_per_step": B,
                "solves_per_s": round(total_structures / elapsed, 3),
                "parallelism": "independent structures sharded over GPUs, no data-path collective",
            },
            "roofline": roofline,
            "phases_ms_profiled_step": phases,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 spends ~12 s in the CPU baseline; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

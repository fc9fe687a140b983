"""
Batched-structure axis: many independent ANM / GNM solves per GPU, sharded over the GPUs of a node.

The reference has no batch axis (one model object = one structure, anm.py:62-63); this module adds
the one parallel axis the path offers — independent structures — in the shape BASELINE.json's
north star asks for: structures are partitioned over ranks (one process per GPU), every rank
assembles and eigendecomposes its shard with the batched device entry points
(``sc_dev_hessian_f64`` / ``sc_dev_eigh_f64``), and RCCL (``torch.distributed`` backend ``nccl``) is
used only to scatter coordinates from / gather eigenvalues to the root rank.  There is no
collective inside the data path.

torch is used strictly as plumbing: device buffers, the stream handed to the C ABI, and the
process group.
"""

import ctypes as C

import numpy as np

from . import _hip
from .forcefield import device_plan

__all__ = ["DeviceBatchSolver", "shard_bounds", "solve_sharded", "partition_lpt", "solve_ragged"]


def shard_bounds(n_items, world_size, rank):
    """Contiguous, balanced partition of ``n_items`` structures: rank r owns [lo, hi)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DeviceBatchSolver:
    """
    ANM (dim=3) or GNM (dim=1) eigensolves for a batch of equally sized structures whose
    coordinates already live in HBM.  Buffers are torch CUDA tensors; all work is enqueued on
    torch's current stream through a ``sc_ctx`` bound to that stream.
    """

    def __init__(self, n_atoms, batch, force_field, dim=3, device=None, want_vectors=True):
        import torch

        self.torch = torch
        self.n_atoms, self.batch, self.dim = int(n_atoms), int(batch), int(dim)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        ff_desc, patch, fused = device_plan(force_field)
        if not fused or patch is not None:
            raise ValueError("the batched device path supports Invariant / Hinsen / ParameterFree force fields")
        self._ff = ff_desc
        self._L = _hip.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx = _hip.Context(self.device.index, stream=stream)
        m = self.n_atoms * self.dim
        self.m = m
        f64 = torch.float64
        self.matrix = torch.empty((self.batch, m, m), dtype=f64, device=self.device)
        self.w = torch.empty((self.batch, m), dtype=f64, device=self.device)
        self.v = torch.empty((self.batch, m, m), dtype=f64, device=self.device) if want_vectors else None

    def set_profiling(self, on):
        self.ctx.check(self._L.sc_ctx_set_profiling(self.ctx.handle, 1 if on else 0))

    def last_timings(self):
        t = (C.c_double * 6)()
        self.ctx.check(self._L.sc_last_eigh_timings(self.ctx.handle, t))
        out = {"tridiag_ms": t[0], "tridiag_eigen_ms": t[1], "backtransform_ms": t[2]}
        if t[5] > 0:   # two-stage tridiagonalisation
            out.update(two_stage=True, band_reduction_ms=t[3], bulge_chasing_ms=t[4], bt2_apply_ms=t[5])
            for name in ("panel_qr", "symm", "syr2k", "dia_tfactor", "dc_gemm", "bt1_w", "bt1_update"):
                ms = C.c_double(0.0)
                if self._L.sc_last_eigh_phase_ms(self.ctx.handle, name.encode(), C.byref(ms)) == 0:
                    out[name + "_ms"] = ms.value
        else:
            out.update(two_stage=False, symv_ms=t[3], syr2k_ms=t[4])
        return out

    def assemble(self, coord):
        """coord: (batch, n_atoms, 3) float64 CUDA tensor -> self.matrix (Hessian / Kirchhoff)."""
        assert coord.is_cuda and coord.dtype == self.torch.float64 and coord.is_contiguous()
        assert tuple(coord.shape) == (self.batch, self.n_atoms, 3)
        fn = self._L.sc_dev_hessian_f64 if self.dim == 3 else self._L.sc_dev_kirchhoff_f64
        self.ctx.check(fn(self.ctx.handle, C.c_void_p(coord.data_ptr()), self.n_atoms, self.batch,
                          C.byref(self._ff), None, C.c_void_p(self.matrix.data_ptr())))
        return self.matrix

    def eigh(self):
        """Eigendecompose self.matrix (destroyed) -> (w, v) tensors; v rows are modes (nma.py:63)."""
        vp = C.c_void_p(self.v.data_ptr()) if self.v is not None else None
        self.ctx.check(self._L.sc_dev_eigh_f64(self.ctx.handle, C.c_void_p(self.matrix.data_ptr()), self.m,
                                               self.batch, C.c_void_p(self.w.data_ptr()), vp))
        return self.w, self.v

    def solve(self, coord):
        """One pass of the hot path over the batch: assembly + full eigensolve, all on device."""
        self.last_coord = coord
        self.assemble(coord)
        return self.eigh()


def solve_sharded(coords, force_field, dim=3, want_vectors=False, group=None, solver_factory=None, solver=None):
    """
    Solve ``len(coords)`` independent structures over all ranks of ``group``.

    ``coords``: (B, n_atoms, 3) float64 ndarray on the ROOT rank (rank 0); other ranks pass None.
    Returns on rank 0 the (B, dim*n_atoms) eigenvalue array (and, if ``want_vectors``, nothing more:
    eigenvectors stay sharded on the ranks that computed them and are returned per rank); on
    other ranks returns that rank's local results.

    Exchange steps (the only communication): scatter of coordinate shards, gather of eigenvalues.
    ``solver_factory(n_atoms, batch)`` may replace the device solver (used by the CPU/gloo tests,
    which check the sharding and the collectives, not the arithmetic).  ``solver``: a
    :class:`DeviceBatchSolver` built for this rank's shard size, reused across calls (its buffers and
    the eigensolver workspace are then allocated once).
    """
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    backend = dist.get_backend(group) if distributed else None
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" or (
        not distributed and torch.cuda.is_available()) else torch.device("cpu")

    # ---- metadata: (B, n_atoms) from the root ------------------------------------------------------
    meta = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == 0:
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        meta[0], meta[1] = coords.shape[0], coords.shape[1]
    if distributed:
        dist.broadcast(meta, src=0, group=group)
    n_items, n_atoms = int(meta[0]), int(meta[1])
    lo, hi = shard_bounds(n_items, world, rank)
    max_local = shard_bounds(n_items, world, 0)[1]

    # ---- scatter coordinates (padded to equal shard sizes: scatter needs uniform shapes) -------------
    local = torch.zeros((max_local, n_atoms, 3), dtype=torch.float64, device=dev)
    if distributed:
        chunks = None
        if rank == 0:
            full = torch.from_numpy(coords).to(dev)
            chunks = []
            for r in range(world):
                a, b = shard_bounds(n_items, world, r)
                c = torch.zeros((max_local, n_atoms, 3), dtype=torch.float64, device=dev)
                c[: b - a] = full[a:b]
                chunks.append(c)
        dist.scatter(local, chunks, src=0, group=group)
    else:
        local[: hi - lo] = torch.from_numpy(coords[lo:hi]).to(dev)

    # ---- local solve ----------------------------------------------------------------------------------
    nloc = hi - lo
    m = n_atoms * dim
    w_local = torch.zeros((max_local, m), dtype=torch.float64, device=dev)
    v_local = None
    if nloc > 0:
        if solver_factory is None:
            if solver is None:
                solver = DeviceBatchSolver(n_atoms, nloc, force_field, dim=dim, want_vectors=want_vectors)
            elif (solver.n_atoms, solver.batch, solver.dim) != (n_atoms, nloc, dim):
                raise ValueError(f"solver was built for {(solver.n_atoms, solver.batch, solver.dim)}, "
                                 f"this rank's shard is {(n_atoms, nloc, dim)}")
            # (with a CPU backend such as gloo the shards travel as CPU tensors: onto the solver's GPU and back)
            w, v_local = solver.solve(local[:nloc].to(solver.device).contiguous())
            w_local[:nloc] = w.to(dev)
        else:
            w_np, v_local = solver_factory(n_atoms, nloc)(local[:nloc].cpu().numpy())
            w_local[:nloc] = torch.from_numpy(np.asarray(w_np)).to(dev)

    # ---- gather eigenvalues on the root -----------------------------------------------------------------
    if distributed:
        gathered = [torch.zeros_like(w_local) for _ in range(world)] if rank == 0 else None
        dist.gather(w_local, gathered, dst=0, group=group)
        if rank == 0:
            out = np.empty((n_items, m))
            for r in range(world):
                a, b = shard_bounds(n_items, world, r)
                out[a:b] = gathered[r][: b - a].cpu().numpy()
            return out, v_local
        return w_local[:nloc].cpu().numpy(), v_local
    return w_local[:nloc].cpu().numpy(), v_local


def partition_lpt(costs, n_bins):
    """
    Longest-processing-time partition (structures of different sizes cost ~ N^3 each): items sorted by cost,
    largest first (ties: lower index first), each one to the currently least loaded bin (ties: lower bin).  Returns
    ``n_bins`` index lists, each in ascending item order.  Deterministic, so every rank computes the same partition
    from the broadcast sizes and no assignment has to be communicated.
    """
    costs = [float(c) for c in costs]
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * n_bins
    bins = [[] for _ in range(n_bins)]
    for i in order:
        b = min(range(n_bins), key=lambda r: (load[r], r))
        bins[b].append(i)
        load[b] += costs[i]
    return [sorted(b) for b in bins]


def solve_ragged(coords_list, force_field, dim=3, group=None, solver_factory=None, solvers=None):
    """
    Independent structures of DIFFERENT sizes over all ranks of ``group``.

    ``coords_list``: list of (N_i, 3) float64 arrays on the ROOT rank (rank 0); other ranks pass None.  The root
    broadcasts the sizes, every rank derives the same :func:`partition_lpt` with cost N_i^3, the root scatters one packed
    (padded) coordinate buffer per rank, each rank solves its structures -- grouped by size, one
    :class:`DeviceBatchSolver` batch per distinct N -- and the root gathers the packed eigenvalues.  Returns on rank 0
    the list of eigenvalue arrays in input order, on the other ranks a dict {item index: eigenvalues} of the local items.
    Two exchange steps, as in :func:`solve_sharded`; eigenvectors are not returned (they stay where they were computed,
    inside the solvers' buffers).  ``solvers``: optional dict {(n_atoms, batch): DeviceBatchSolver} reused across calls.
    """
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    backend = dist.get_backend(group) if distributed else None
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" or (
        not distributed and torch.cuda.is_available() and solver_factory is None) else torch.device("cpu")

    # ---- sizes from the root ---------------------------------------------------------------------------
    n_items = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        coords_list = [np.ascontiguousarray(c, dtype=np.float64) for c in coords_list]
        for c in coords_list:
            if c.ndim != 2 or c.shape[1] != 3:
                raise ValueError(f"every structure must be an (N, 3) coordinate array, got {c.shape}")
        n_items[0] = len(coords_list)
    if distributed:
        dist.broadcast(n_items, src=0, group=group)
    B = int(n_items[0])
    sizes = torch.zeros(max(B, 1), dtype=torch.int64, device=dev)
    if rank == 0 and B:
        sizes[:B] = torch.tensor([c.shape[0] for c in coords_list], dtype=torch.int64)
    if distributed:
        dist.broadcast(sizes, src=0, group=group)
    sizes = [int(x) for x in sizes[:B].cpu()]
    parts = partition_lpt([float(n) ** 3 for n in sizes], world)
    mine = parts[rank]
    atoms_per_rank = [sum(sizes[i] for i in part) for part in parts]
    pad_atoms = max(atoms_per_rank + [1])

    # ---- scatter the packed coordinates (one padded buffer per rank) -----------------------------------------
    local = torch.zeros((pad_atoms, 3), dtype=torch.float64, device=dev)
    if distributed:
        chunks = None
        if rank == 0:
            chunks = []
            for part in parts:
                c = torch.zeros((pad_atoms, 3), dtype=torch.float64, device=dev)
                off = 0
                for i in part:
                    c[off: off + sizes[i]] = torch.from_numpy(coords_list[i]).to(dev)
                    off += sizes[i]
                chunks.append(c)
        dist.scatter(local, chunks, src=0, group=group)
    else:
        off = 0
        for i in mine:
            local[off: off + sizes[i]] = torch.from_numpy(coords_list[i]).to(dev)
            off += sizes[i]

    # ---- local solves, one batch per distinct size --------------------------------------------------------------
    offsets, off = {}, 0
    for i in mine:
        offsets[i] = off
        off += sizes[i]
    w_local = torch.zeros(pad_atoms * dim, dtype=torch.float64, device=dev)
    results = {}
    for n_atoms in sorted({sizes[i] for i in mine}):
        items = [i for i in mine if sizes[i] == n_atoms]
        batch = torch.stack([local[offsets[i]: offsets[i] + n_atoms] for i in items]).contiguous()
        if solver_factory is not None:
            w_np, _ = solver_factory(n_atoms, len(items))(batch.cpu().numpy())
            w = torch.from_numpy(np.asarray(w_np))
        else:
            key = (n_atoms, len(items))
            solver = solvers.get(key) if solvers is not None else None
            if solver is None:
                solver = DeviceBatchSolver(n_atoms, len(items), force_field, dim=dim, want_vectors=False)
                if solvers is not None:
                    solvers[key] = solver
            w, _ = solver.solve(batch.to(solver.device))
        for k, i in enumerate(items):
            w_local[offsets[i] * dim: (offsets[i] + n_atoms) * dim] = w[k].to(dev)
            results[i] = w[k].cpu().numpy().copy()

    # ---- gather the packed eigenvalues ------------------------------------------------------------------------------
    if distributed:
        gathered = [torch.zeros_like(w_local) for _ in range(world)] if rank == 0 else None
        dist.gather(w_local, gathered, dst=0, group=group)
        if rank != 0:
            return results
        out = [None] * B
        for r, part in enumerate(parts):
            off = 0
            buf = gathered[r].cpu().numpy()
            for i in part:
                out[i] = buf[off * dim: (off + sizes[i]) * dim].copy()
                off += sizes[i]
        return out
    return [results[i] for i in range(B)]

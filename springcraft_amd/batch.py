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

__all__ = ["DeviceBatchSolver", "RaggedBatchSolver", "shard_bounds", "solve_sharded", "partition_lpt", "size_buckets",
           "solve_ragged"]


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

    ``masses``: None, or (batch, n_atoms) / (n_atoms,) atomic masses: the matrices are mass-weighted as
    ``ANM.hessian`` / ``GNM.kirchhoff`` do (anm.py:89-94,112-113; gnm.py:85-87,104-105).
    ``subset_by_index=(lo, hi)``: only the eigenpairs with ascending index lo..hi (inclusive) through the
    partial-spectrum path (no reference counterpart; BASELINE config 5): ``w`` is (batch, m), ``v`` (batch, m, n).
    Structures of different sizes, patched or tabulated force fields: :class:`RaggedBatchSolver`.
    """

    def __init__(self, n_atoms, batch, force_field, dim=3, device=None, want_vectors=True, masses=None,
                 subset_by_index=None):
        import torch

        self.torch = torch
        self.n_atoms, self.batch, self.dim = int(n_atoms), int(batch), int(dim)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        ff_desc, patch, fused = device_plan(force_field)
        if not fused or patch is not None or ff_desc.kind == _hip.SC_FF_TABULATED:
            raise ValueError("DeviceBatchSolver takes Invariant / Hinsen / ParameterFree force fields; use "
                             "RaggedBatchSolver for patched or tabulated ones")
        self._ff = ff_desc
        self._L = _hip.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx = _hip.Context(self.device.index, stream=stream)
        m = self.n_atoms * self.dim
        self.m = m
        f64 = torch.float64
        self.subset = None
        nvec = m
        if subset_by_index is not None:
            lo, hi = int(subset_by_index[0]), int(subset_by_index[1])
            if not 0 <= lo <= hi < m:
                raise ValueError(f"subset_by_index {subset_by_index} outside 0..{m - 1}")
            self.subset = (lo, hi)
            nvec = hi - lo + 1
        self.matrix = torch.empty((self.batch, m, m), dtype=f64, device=self.device)
        self.w = torch.empty((self.batch, nvec), dtype=f64, device=self.device)
        self.v = torch.empty((self.batch, nvec, m), dtype=f64, device=self.device) if want_vectors else None
        self.inv_sqrt_mass = None
        if masses is not None:
            mm = torch.as_tensor(np.asarray(masses, dtype=np.float64) if not torch.is_tensor(masses) else masses,
                                 dtype=f64, device=self.device)
            if mm.ndim == 1:
                mm = mm[None, :].expand(self.batch, -1)
            if tuple(mm.shape) != (self.batch, self.n_atoms):
                raise IndexError(f"{tuple(mm.shape)} masses for a batch of {self.batch} x {self.n_atoms} atoms")
            if bool((mm == 0).any()):
                raise ValueError("masses must not be 0")          # anm.py:85-86
            self.inv_sqrt_mass = (1.0 / torch.sqrt(mm)).contiguous()

    def set_profiling(self, on):
        self.ctx.check(self._L.sc_ctx_set_profiling(self.ctx.handle, 1 if on else 0))

    def last_timings(self):
        return _last_timings(self.ctx, self._L)

    def assemble(self, coord):
        """coord: (batch, n_atoms, 3) float64 CUDA tensor -> self.matrix (Hessian / Kirchhoff)."""
        assert coord.is_cuda and coord.dtype == self.torch.float64 and coord.is_contiguous()
        assert tuple(coord.shape) == (self.batch, self.n_atoms, 3)
        fn = self._L.sc_dev_hessian_f64 if self.dim == 3 else self._L.sc_dev_kirchhoff_f64
        wp = C.c_void_p(self.inv_sqrt_mass.data_ptr()) if self.inv_sqrt_mass is not None else None
        self.ctx.check(fn(self.ctx.handle, C.c_void_p(coord.data_ptr()), self.n_atoms, self.batch,
                          C.byref(self._ff), wp, C.c_void_p(self.matrix.data_ptr())))
        return self.matrix

    def eigh(self):
        """Eigendecompose self.matrix (destroyed) -> (w, v) tensors; v rows are modes (nma.py:63)."""
        vp = C.c_void_p(self.v.data_ptr()) if self.v is not None else None
        if self.subset is None:
            self.ctx.check(self._L.sc_dev_eigh_f64(self.ctx.handle, C.c_void_p(self.matrix.data_ptr()), self.m,
                                                   self.batch, C.c_void_p(self.w.data_ptr()), vp))
        else:
            self.ctx.check(self._L.sc_dev_eigh_range_f64(self.ctx.handle, C.c_void_p(self.matrix.data_ptr()), self.m,
                                                         self.batch, self.subset[0], self.subset[1],
                                                         C.c_void_p(self.w.data_ptr()), vp))
        return self.w, self.v

    def solve(self, coord):
        """
        One pass of the hot path over the batch: assembly + eigensolve, all on device.  Only ENQUEUES (the result tensors
        are valid in stream order); call :meth:`finish` before trusting them on the host.
        """
        self.assemble(coord)
        return self.eigh()

    def finish(self):
        """
        Wait for the solves enqueued so far and raise what they could only find out on the device:
        ``np.linalg.LinAlgError`` -- what ``np.linalg.eigh`` raises at nma.py:61 -- if a matrix held a NaN / Inf entry
        (its eigenvalues come back NaN, the other structures of the batch are unaffected) or a tridiagonal QL iteration
        did not converge.  The condition is reported once.  Returns (w, v).
        """
        self.ctx.synchronize()
        return self.w, self.v


class RaggedBatchSolver:
    """
    ONE batched solve for structures that differ: in size, in force field -- any built-in one,
    :class:`TabulatedForceField` (forcefield.py:369-533) and :class:`PatchedForceField` (forcefield.py:117-261) around
    those included -- and in their masses (anm.py:89-94).  The reference models one arbitrary structure per object
    (anm.py:62-63); here every structure gets a slot of one common matrix order in a single batched eigensolve
    (``sc_batch_plan_*`` in the C ABI, which documents the exact padding of the slots).

    User-defined :class:`ForceField` subclasses (``force_constant()`` in Python: doc/advanced.rst:23-70,
    tests/test_interaction.py:92-116) are batched too: as soon as one member needs the host callback, the pair lists of
    ALL structures come back from one device launch (``sc_batch_plan_pairs``), every structure's force field evaluates
    its constants on its own ordered pairs exactly as interaction.py:49 / :96 do, and one device pass fills all padded
    slots (``sc_batch_plan_fill_from_pairs_f64``; asymmetric constants honoured as interaction.py:50-52,103-104).

    sizes         atom counts, one per structure
    force_fields  one force field for all structures (only if it is not bound to particular atoms) or one per structure
    masses        None, or one entry per structure: None or an (n_atoms,) array
    order         common matrix order, default dim * max(sizes)

    ``solve(coord)`` takes the structures' coordinates back to back, (sum(sizes), 3) float64 on the device, and returns
    the padded result tensors (w (B, order), v (B, order, order)); ``results()`` slices them into per-structure views
    (w_i (dim n_i,), v_i (dim n_i, dim n_i), rows = modes as nma.py:63).
    """

    def __init__(self, sizes, force_fields, dim=3, masses=None, device=None, want_vectors=True, order=None):
        import torch

        self.torch = torch
        self.sizes = [int(n) for n in sizes]
        self.batch, self.dim = len(self.sizes), int(dim)
        if self.batch == 0:
            raise ValueError("no structures")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        if not isinstance(force_fields, (list, tuple)):
            force_fields = [force_fields] * self.batch
        if len(force_fields) != self.batch:
            raise ValueError(f"{len(force_fields)} force fields for {self.batch} structures")
        self._L = _hip.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx = _hip.Context(self.device.index, stream=stream)
        self._keep = []          # descriptor memory must outlive sc_batch_plan_create
        descs = (_hip.StructureDesc * self.batch)()
        plans = {}
        self.force_fields = list(force_fields)
        self.host_callback = False       # True: gamma comes from force_constant() in Python for the whole batch
        for b, (n, ff) in enumerate(zip(self.sizes, force_fields)):
            if ff.natoms is not None and ff.natoms != n:
                raise ValueError(f"structure {b}: the force field was built for {ff.natoms} atoms, the structure has {n}")
            if id(ff) not in plans:
                ff_desc, patch, fused = device_plan(ff)
                if not fused:
                    self.host_callback = True
                pd = None
                if patch is not None:
                    from .interaction import _normalised_patch

                    # (negative indices, boolean masks, IndexError / self-pair ValueError as compute_* raise them)
                    pd = _normalised_patch(patch, n, self._keep)
                plans[id(ff)] = (ff_desc, pd)
                self._keep += [ff_desc, pd, ff]
            ff_desc, pd = plans[id(ff)]
            descs[b].n_atoms = n
            descs[b].ff = C.pointer(ff_desc)
            descs[b].patch = C.pointer(pd) if pd is not None else None
        h = C.c_void_p()
        self.ctx.check(self._L.sc_batch_plan_create(self.ctx.handle, self.dim, descs, self.batch,
                                                    0 if order is None else int(order), C.byref(h)))
        self._plan = h
        self.order = int(self._L.sc_batch_plan_order(h))
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        f64 = torch.float64
        m = self.order
        self.matrix = torch.empty((self.batch, m, m), dtype=f64, device=self.device)
        self.w = torch.empty((self.batch, m), dtype=f64, device=self.device)
        self.v = torch.empty((self.batch, m, m), dtype=f64, device=self.device) if want_vectors else None
        self.inv_sqrt_mass = None
        if masses is not None:
            if len(masses) != self.batch:
                raise IndexError(f"{len(masses)} mass entries for {self.batch} structures")
            packed = np.ones(int(self.offsets[-1]))
            for b, mb in enumerate(masses):
                if mb is None:
                    continue
                mb = np.asarray(mb, dtype=np.float64)
                if mb.shape != (self.sizes[b],):
                    raise IndexError(f"structure {b}: {mb.shape} masses for {self.sizes[b]} atoms")   # anm.py:81-84
                if np.any(mb == 0):
                    raise ValueError("masses must not be 0")                                            # anm.py:85-86
                packed[self.offsets[b]: self.offsets[b + 1]] = 1.0 / np.sqrt(mb)
            self.inv_sqrt_mass = torch.from_numpy(packed).to(self.device)

    def set_profiling(self, on):
        self.ctx.check(self._L.sc_ctx_set_profiling(self.ctx.handle, 1 if on else 0))

    def last_timings(self):
        return _last_timings(self.ctx, self._L)

    def assemble(self, coord):
        """coord: (sum(sizes), 3) float64 CUDA tensor -> self.matrix, one padded slot per structure."""
        assert coord.is_cuda and coord.dtype == self.torch.float64 and coord.is_contiguous()
        assert tuple(coord.shape) == (int(self.offsets[-1]), 3)
        wp = C.c_void_p(self.inv_sqrt_mass.data_ptr()) if self.inv_sqrt_mass is not None else None
        if self.host_callback:
            return self._assemble_from_pairs(coord, wp)
        self.ctx.check(self._L.sc_batch_plan_assemble_f64(self._plan, C.c_void_p(coord.data_ptr()), wp,
                                                          C.c_void_p(self.matrix.data_ptr())))
        return self.matrix

    def pairs(self, coord, want_sq_dist=True):
        """
        Ordered pair lists of all structures from ONE device launch: [(pairs_b (k_b, 2) int64, sq_dist_b (k_b,)), ...],
        per structure what ``compute_kirchhoff`` / ``compute_hessian`` return as ``pairs`` (interaction.py:177-178).
        """
        self.torch.cuda.current_stream(self.device).synchronize()     # coord may come from torch's stream
        cp = C.c_void_p(coord.data_ptr())
        counts = np.zeros(self.batch, dtype=np.int64)
        self.ctx.check(self._L.sc_batch_plan_contacts(self._plan, cp, _hip.ptr(counts)))
        k = int(counts.sum())
        pairs = np.empty((k, 2), dtype=np.int64)
        sq = np.empty(k, dtype=np.float64) if want_sq_dist else None
        off = np.zeros(self.batch + 1, dtype=np.int64)
        self.ctx.check(self._L.sc_batch_plan_pairs(self._plan, cp, k, _hip.ptr(pairs), _hip.ptr(sq), _hip.ptr(off)))
        self._pairs_flat, self._pair_off = pairs, off
        return [(pairs[off[b]: off[b + 1]], sq[off[b]: off[b + 1]] if sq is not None else None) for b in range(self.batch)]

    def _assemble_from_pairs(self, coord, wp):
        per = self.pairs(coord, want_sq_dist=True)
        gamma = np.empty(len(self._pairs_flat), dtype=np.float64)
        for b, ((pb, sqb), ff) in enumerate(zip(per, self.force_fields)):
            g = np.asarray(ff.force_constant(pb[:, 0], pb[:, 1], sqb))   # interaction.py:49,96
            if g.shape != (len(pb),):
                raise ValueError(f"structure {b}: force_constant() returned shape {g.shape} for {len(pb)} pairs")
            gamma[self._pair_off[b]: self._pair_off[b + 1]] = g          # (Tabulated returns float32: promoted here)
        self.ctx.check(self._L.sc_batch_plan_fill_from_pairs_f64(
            self._plan, C.c_void_p(coord.data_ptr()), _hip.ptr(self._pairs_flat), _hip.ptr(self._pair_off),
            _hip.ptr(gamma), wp, C.c_void_p(self.matrix.data_ptr())))
        return self.matrix

    def eigh(self):
        vp = C.c_void_p(self.v.data_ptr()) if self.v is not None else None
        self.ctx.check(self._L.sc_dev_eigh_f64(self.ctx.handle, C.c_void_p(self.matrix.data_ptr()), self.order,
                                               self.batch, C.c_void_p(self.w.data_ptr()), vp))
        return self.w, self.v

    def solve(self, coord):
        self.assemble(coord)
        return self.eigh()

    def finish(self):
        """As :meth:`DeviceBatchSolver.finish`: wait, and raise ``np.linalg.LinAlgError`` for NaN / Inf input or a QL failure."""
        self.ctx.synchronize()
        return self.w, self.v

    def results(self):
        """
        Per-structure views of the last solve: [(w_i, v_i or None), ...].  Waits for the solve and raises
        ``np.linalg.LinAlgError`` as ``np.linalg.eigh`` does at nma.py:61 (see :meth:`finish`).
        """
        self.ctx.synchronize()
        out = []
        for b, n in enumerate(self.sizes):
            m = self.dim * n
            out.append((self.w[b, :m], self.v[b, :m, :m] if self.v is not None else None))
        return out

    def close(self):
        if getattr(self, "_plan", None) is not None and self._plan.value:
            self._L.sc_batch_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def size_buckets(sizes, max_flop_ratio=1.25):
    """
    Groups structures whose sizes are close enough to share one padded batch: sorted by size, largest first, a bucket
    takes every structure with (N_max / N)^3 <= max_flop_ratio (the eigensolve costs ~ order^3, so no member pays more
    than that factor for its padding).  Returns lists of item indices, each in ascending item order; deterministic.
    """
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    buckets, cur, n_max = [], [], None
    for i in order:
        if cur and (n_max / sizes[i]) ** 3 > max_flop_ratio:
            buckets.append(sorted(cur))
            cur = []
        if not cur:
            n_max = sizes[i]
        cur.append(i)
    if cur:
        buckets.append(sorted(cur))
    return buckets


def _last_timings(ctx, L):
    """Phase durations (ms) of the context's most recent profiled eigensolve as a dict."""
    t = (C.c_double * 6)()
    ctx.check(L.sc_last_eigh_timings(ctx.handle, t))
    out = {"tridiag_ms": t[0], "tridiag_eigen_ms": t[1], "backtransform_ms": t[2]}
    if t[5] > 0:   # two-stage tridiagonalisation
        out.update(two_stage=True, band_reduction_ms=t[3], bulge_chasing_ms=t[4], bt2_apply_ms=t[5])
    else:
        out.update(two_stage=False, symv_ms=t[3], syr2k_ms=t[4])
    for name in ("resident_tridiag", "panel_qr", "symm", "syr2k", "dia_tfactor", "dc_gemm", "bt1_w", "bt1_update", "sturm",
                 "stein", "cholqr"):
        ms = C.c_double(0.0)
        if L.sc_last_eigh_phase_ms(ctx.handle, name.encode(), C.byref(ms)) == 0 and name + "_ms" not in out:
            out[name + "_ms"] = ms.value
    # (not a time: 1e9 flops of the D&C merge GEMMs, summed on the device from their records -- the sizes depend on deflation)
    g = C.c_double(0.0)
    if L.sc_last_eigh_phase_ms(ctx.handle, b"dc_gemm_gflop", C.byref(g)) == 0:
        out["dc_gemm_gflop"] = g.value
    return out


def solve_sharded(coords, force_field, dim=3, want_vectors=False, group=None, solver_factory=None, solver=None):
    """
    Solve ``len(coords)`` independent structures over all ranks of ``group``.

    ``coords``: (B, n_atoms, 3) float64 ndarray on the ROOT rank (rank 0); other ranks pass None.
    Returns on rank 0 the (B, dim*n_atoms) eigenvalue array (and, if ``want_vectors``, nothing more:
    eigenvectors stay sharded on the ranks that computed them and are returned per rank); on
    other ranks returns that rank's local results.

    Exchange steps (the only communication): scatter of coordinate shards, gather of eigenvalues.
    A solve that fails on the device (NaN / Inf in a matrix, QL iteration without convergence) raises
    ``np.linalg.LinAlgError`` -- what ``np.linalg.eigh`` raises at nma.py:61 -- on the rank it happened on AND on the
    root, both after the gather (the flag travels with the eigenvalues, so no rank is left waiting in a collective).
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
    # (row max_local of the gathered buffer carries the rank's deferred solver status: 0 = fine)
    w_local = torch.zeros((max_local + 1, m), dtype=torch.float64, device=dev)
    v_local = None
    failure = None
    if nloc > 0:
        if solver_factory is None:
            if solver is None:
                solver = DeviceBatchSolver(n_atoms, nloc, force_field, dim=dim, want_vectors=want_vectors)
            elif (solver.n_atoms, solver.batch, solver.dim) != (n_atoms, nloc, dim):
                raise ValueError(f"solver was built for {(solver.n_atoms, solver.batch, solver.dim)}, "
                                 f"this rank's shard is {(n_atoms, nloc, dim)}")
            # (with a CPU backend such as gloo the shards travel as CPU tensors: onto the solver's GPU and back)
            w, v_local = solver.solve(local[:nloc].to(solver.device).contiguous())
            try:
                solver.finish()
            except np.linalg.LinAlgError as e:     # keep the collectives matched: report after the gather
                failure = e
            w_local[:nloc] = w.to(dev)
        else:
            try:
                w_np, v_local = solver_factory(n_atoms, nloc)(local[:nloc].cpu().numpy())
                w_local[:nloc] = torch.from_numpy(np.asarray(w_np)).to(dev)
            except np.linalg.LinAlgError as e:
                failure = e
                w_local[:nloc] = float("nan")
    if failure is not None:
        w_local[max_local, 0] = 1.0

    # ---- gather eigenvalues on the root -----------------------------------------------------------------
    if distributed:
        gathered = [torch.zeros_like(w_local) for _ in range(world)] if rank == 0 else None
        dist.gather(w_local, gathered, dst=0, group=group)
        if failure is not None:
            raise np.linalg.LinAlgError(f"rank {rank}: {failure}")
        if rank == 0:
            out = np.empty((n_items, m))
            failed = []
            for r in range(world):
                a, b = shard_bounds(n_items, world, r)
                g = gathered[r].cpu().numpy()
                out[a:b] = g[: b - a]
                if g[max_local, 0] != 0.0:
                    failed.append(r)
            if failed:
                raise np.linalg.LinAlgError(f"Eigenvalues did not converge on rank(s) {failed} (NaN / Inf input or a "
                                            "failed QL iteration; see that rank's exception)")
            return out, v_local
        return w_local[:nloc].cpu().numpy(), v_local
    if failure is not None:
        raise failure
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


def solve_ragged(coords_list, force_field, dim=3, group=None, solver_factory=None, solvers=None, max_flop_ratio=1.25):
    """
    Independent structures of DIFFERENT sizes over all ranks of ``group``.

    ``coords_list``: list of (N_i, 3) float64 arrays on the ROOT rank (rank 0); other ranks pass None.  The root
    broadcasts the sizes, every rank derives the same :func:`partition_lpt` with cost N_i^3, the root scatters one packed
    (padded) coordinate buffer per rank, each rank solves its structures -- ONE padded :class:`RaggedBatchSolver` batch
    per bucket of similar sizes (:func:`size_buckets`: (N_max / N)^3 <= ``max_flop_ratio``, so no member pays more than
    that factor for its padding) -- and the root gathers the packed eigenvalues.  Returns on rank 0
    the list of eigenvalue arrays in input order, on the other ranks a dict {item index: eigenvalues} of the local items.
    Two exchange steps, as in :func:`solve_sharded`; eigenvectors are not returned (they stay where they were computed,
    inside the solvers' buffers).  ``solvers``: optional dict {tuple of the bucket's sizes: RaggedBatchSolver} reused
    across calls.  Device-side failures raise ``np.linalg.LinAlgError`` after the gather, as in :func:`solve_sharded`.
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

    # ---- local solves: one padded batch per bucket of similar sizes (injected CPU solvers: one per distinct size) ------
    offsets, off = {}, 0
    for i in mine:
        offsets[i] = off
        off += sizes[i]
    # (the last element of the gathered buffer carries the rank's deferred solver status: 0 = fine)
    w_local = torch.zeros(pad_atoms * dim + 1, dtype=torch.float64, device=dev)
    results = {}
    failure = None
    if solver_factory is not None:
        for n_atoms in sorted({sizes[i] for i in mine}):
            items = [i for i in mine if sizes[i] == n_atoms]
            batch = torch.stack([local[offsets[i]: offsets[i] + n_atoms] for i in items]).contiguous()
            try:
                w_np, _ = solver_factory(n_atoms, len(items))(batch.cpu().numpy())
            except np.linalg.LinAlgError as e:
                failure = e
                w_np = np.full((len(items), n_atoms * dim), np.nan)
            w = torch.from_numpy(np.asarray(w_np))
            for k, i in enumerate(items):
                w_local[offsets[i] * dim: (offsets[i] + n_atoms) * dim] = w[k].to(dev)
                results[i] = w[k].cpu().numpy().copy()
    else:
        for bucket in size_buckets([sizes[i] for i in mine], max_flop_ratio):
            items = [mine[k] for k in bucket]
            key = tuple(sizes[i] for i in items)
            solver = solvers.get(key) if solvers is not None else None
            if solver is None:
                solver = RaggedBatchSolver(key, force_field, dim=dim, want_vectors=False)
                if solvers is not None:
                    solvers[key] = solver
            packed = torch.cat([local[offsets[i]: offsets[i] + sizes[i]] for i in items]).to(solver.device).contiguous()
            solver.solve(packed)
            try:
                per_structure = solver.results()       # waits, raises what the device found
            except np.linalg.LinAlgError as e:         # keep the collectives matched: report after the gather
                failure = e
                per_structure = solver.results()
            for i, (wi, _) in zip(items, per_structure):
                w_local[offsets[i] * dim: (offsets[i] + sizes[i]) * dim] = wi.to(dev)
                results[i] = wi.cpu().numpy().copy()
    if failure is not None:
        w_local[-1] = 1.0

    # ---- gather the packed eigenvalues ------------------------------------------------------------------------------
    if distributed:
        gathered = [torch.zeros_like(w_local) for _ in range(world)] if rank == 0 else None
        dist.gather(w_local, gathered, dst=0, group=group)
        if failure is not None:
            raise np.linalg.LinAlgError(f"rank {rank}: {failure}")
        if rank != 0:
            return results
        out = [None] * B
        failed = []
        for r, part in enumerate(parts):
            off = 0
            buf = gathered[r].cpu().numpy()
            if buf[-1] != 0.0:
                failed.append(r)
            for i in part:
                out[i] = buf[off * dim: (off + sizes[i]) * dim].copy()
                off += sizes[i]
        if failed:
            raise np.linalg.LinAlgError(f"Eigenvalues did not converge on rank(s) {failed} (NaN / Inf input or a failed "
                                        "QL iteration; see that rank's exception)")
        return out
    if failure is not None:
        raise failure
    return [results[i] for i in range(B)]

"""
world_size-2 gloo test of the N>1 path (solve_sharded): the batch scatter, the per-rank shard and the
eigenvalue gather, on CPU.  The device solver is replaced by the oracle through ``solver_factory``
(this is a test: it checks the sharding / collectives, not the arithmetic).
"""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, queue):
    import torch.distributed as dist

    from oracle import enm_oracle as orc
    from springcraft_amd.batch import solve_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def factory(n_atoms, batch):
        def run(coords):
            ws = []
            for c in coords:
                h, _ = orc.compute_hessian(c, orc.invariant_ff(8.0))
                ws.append(orc.eigen(h)[0])
            return np.array(ws), None
        return run

    coords = None
    if rank == 0:
        coords = np.stack([orc.synthetic_coord(30, s, 12.0) for s in range(n_items)])
    w, _ = solve_sharded(coords, None, dim=3, solver_factory=factory)
    if rank == 0:
        queue.put(w)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 2, 1])
def test_solve_sharded_two_ranks(n_items):
    import torch.multiprocessing as mp

    from oracle import enm_oracle as orc

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    w = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert w.shape == (n_items, 90)
    for s in range(n_items):
        h, _ = orc.compute_hessian(orc.synthetic_coord(30, s, 12.0), orc.invariant_ff(8.0))
        assert np.allclose(w[s], orc.eigen(h)[0], atol=1e-10)


def _failing_worker(rank, world, port, queue):
    import torch.distributed as dist

    from oracle import enm_oracle as orc
    from springcraft_amd.batch import solve_ragged, solve_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def factory(n_atoms, batch):
        def run(coords):
            ws = []
            for c in coords:
                h, _ = orc.compute_hessian(c, orc.hinsen_ff())      # no cutoff: a NaN coordinate reaches the matrix
                if not np.isfinite(h).all():                        # what the device solver reports (LAPACK builds differ:
                    raise np.linalg.LinAlgError("Eigenvalues did not converge")   # some raise at nma.py:61, some return NaN)
                ws.append(orc.eigen(h)[0])
            return np.array(ws), None
        return run

    seen = []
    coords = None
    if rank == 0:
        coords = np.stack([orc.synthetic_coord(20, s, 12.0) for s in range(4)])
        coords[3, 5, 1] = np.nan                                    # structure 3 is solved by rank 1
    try:
        solve_sharded(coords, None, dim=3, solver_factory=factory)
        seen.append("sharded: no error")
    except np.linalg.LinAlgError as e:
        seen.append("sharded: " + str(e))
    try:
        solve_ragged(list(coords) if rank == 0 else None, None, dim=3, solver_factory=factory)
        seen.append("ragged: no error")
    except np.linalg.LinAlgError as e:
        seen.append("ragged: " + str(e))
    # both calls left the collectives matched: one more collective still works
    dist.barrier()
    queue.put((rank, seen))
    dist.destroy_process_group()


def test_device_side_failure_raises_linalgerror_on_root_and_owner_without_hanging():
    """
    np.linalg.eigh raises LinAlgError for non-finite input (nma.py:61).  In the sharded form the failing rank must not
    skip the gather (the others would wait for ever): the flag travels with the eigenvalues, the owner and the root
    raise after the exchange.
    """
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0].startswith("sharded: Eigenvalues did not converge on rank(s) [1]")
    assert got[1][0].startswith("sharded: rank 1:")
    # (solve_ragged's LPT partition decides who owns the NaN structure: the root reports the owner's rank)
    assert got[0][1].startswith("ragged: ") and "no error" not in got[0][1]
    assert any("rank" in got[r][1] and "no error" not in got[r][1] for r in (0, 1))


# ---- ragged batches: longest-processing-time partition by N^3 -------------------------------------------
def test_partition_lpt_properties():
    from springcraft_amd.batch import partition_lpt

    sizes = [40, 12, 25, 12, 30, 12, 7, 33]
    costs = [float(n) ** 3 for n in sizes]
    for world in (1, 2, 3, 8, 11):
        parts = partition_lpt(costs, world)
        assert len(parts) == world
        assert sorted(i for p in parts for i in p) == list(range(len(sizes)))   # every item exactly once
        assert parts == partition_lpt(costs, world)                               # deterministic
        load = [sum(costs[i] for i in p) for p in parts]
        # Graham's bound for LPT: makespan <= (4/3 - 1/(3m)) OPT, and OPT >= max(largest item, mean load)
        opt_lb = max(max(costs), sum(costs) / world)
        assert max(load) <= (4.0 / 3.0 - 1.0 / (3.0 * world)) * opt_lb * (1 + 1e-12) or max(load) == max(costs)
    # the two largest items never share a bin when there are at least two bins
    parts = partition_lpt(costs, 2)
    assert (0 in parts[0]) != (7 in parts[0])
    assert partition_lpt([], 3) == [[], [], []]


_RAGGED = [30, 12, 25, 12, 40, 12, 7]


def _ragged_worker(rank, world, port, queue):
    import torch.distributed as dist

    from oracle import enm_oracle as orc
    from springcraft_amd.batch import solve_ragged

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []

    def factory(n_atoms, batch):
        seen.append((n_atoms, batch))

        def run(coords):
            assert coords.shape == (batch, n_atoms, 3)
            ws = [orc.eigen(orc.compute_hessian(c, orc.invariant_ff(8.0))[0])[0] for c in coords]
            return np.array(ws), None
        return run

    coords = None
    if rank == 0:
        coords = [orc.synthetic_coord(n, s, 12.0) for s, n in enumerate(_RAGGED)]
    out = solve_ragged(coords, None, dim=3, solver_factory=factory)
    queue.put((rank, out, seen))
    dist.destroy_process_group()


def test_solve_ragged_two_ranks():
    import torch.multiprocessing as mp

    from oracle import enm_oracle as orc
    from springcraft_amd.batch import partition_lpt

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, out, seen = q.get(timeout=120)
        got[r] = (out, seen)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = got[0][0]
    assert len(w) == len(_RAGGED)
    for s, n in enumerate(_RAGGED):
        h, _ = orc.compute_hessian(orc.synthetic_coord(n, s, 12.0), orc.invariant_ff(8.0))
        assert w[s].shape == (3 * n,)
        assert np.allclose(w[s], orc.eigen(h)[0], atol=1e-10)
    # every rank solved exactly its LPT share, one batch per distinct size; the non-root rank returns its items
    parts = partition_lpt([float(n) ** 3 for n in _RAGGED], 2)
    for r in range(2):
        expect = sorted((n, sum(1 for i in parts[r] if _RAGGED[i] == n)) for n in {_RAGGED[i] for i in parts[r]})
        assert sorted(got[r][1]) == expect
    assert sorted(got[1][0].keys()) == parts[1]


def test_solve_ragged_single_process():
    from oracle import enm_oracle as orc
    from springcraft_amd.batch import solve_ragged

    def factory(n_atoms, batch):
        return lambda coords: (np.array([orc.eigen(orc.compute_hessian(c, orc.invariant_ff(8.0))[0])[0] for c in coords]), None)

    coords = [orc.synthetic_coord(n, s, 12.0) for s, n in enumerate([9, 14, 9])]
    w = solve_ragged(coords, None, solver_factory=factory)
    for c, wi in zip(coords, w):
        assert np.allclose(wi, orc.eigen(orc.compute_hessian(c, orc.invariant_ff(8.0))[0])[0], atol=1e-10)
    with pytest.raises(ValueError):
        solve_ragged([np.zeros((4, 2))], None, solver_factory=factory)
    assert solve_ragged([], None, solver_factory=factory) == []


def test_size_buckets_properties():
    """Buckets of similar sizes for the padded batches: a partition, deterministic, no member pays more than the ratio."""
    from springcraft_amd.batch import size_buckets

    rs = np.random.RandomState(0)
    sizes = [int(x) for x in rs.randint(50, 2000, 200)]
    for ratio in (1.0, 1.25, 2.0):
        buckets = size_buckets(sizes, ratio)
        flat = sorted(i for b in buckets for i in b)
        assert flat == list(range(len(sizes)))                       # every structure exactly once
        assert buckets == size_buckets(sizes, ratio)                 # deterministic
        for b in buckets:
            assert b == sorted(b)
            n_max = max(sizes[i] for i in b)
            assert all((n_max / sizes[i]) ** 3 <= ratio + 1e-12 for i in b)
    assert all(len({sizes[i] for i in b}) == 1 for b in size_buckets(sizes, 1.0))    # ratio 1: one bucket per size
    assert size_buckets([], 1.25) == [] and size_buckets([7], 1.25) == [[0]]


def test_bench_numa_binding_is_best_effort():
    """bench.py pins a rank to its GPU's NUMA node from sysfs; on a box without GPUs it must say so and change nothing."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    before = os.sched_getaffinity(0)
    msg = bench.bind_rank_to_numa_node(0)
    assert isinstance(msg, str) and msg
    if msg.startswith("unbound"):
        assert os.sched_getaffinity(0) == before
    else:
        assert os.sched_getaffinity(0) <= before
        os.sched_setaffinity(0, before)
    assert bench.bind_rank_to_numa_node(10 ** 6).startswith("unbound")

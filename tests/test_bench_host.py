"""
bench.py's CPU legs under a multi-rank launch ("next" item 2).

torch.distributed.run exports OMP_NUM_THREADS=1 to every rank when the variable is unset and there is more than one
rank, and bench.py narrows a rank's affinity mask to one NUMA node for the GPU-driving phase.  SURVEY.md section 8(d)
wants the CPU baseline "on the same box's host cores": the spawn environment must carry an explicit thread count and
the baseline leg must run on the cores (and BLAS threads) the process started with.  No GPU is needed for any of this.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_spawn_env_carries_the_thread_count(monkeypatch):
    bench = _bench()
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    env = bench.spawn_env(2)
    ncores = len(os.sched_getaffinity(0))
    assert env["OMP_NUM_THREADS"] == str(max(1, ncores // 2))
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a value the user chose is left alone
    monkeypatch.setenv("OMP_NUM_THREADS", "3")
    assert bench.spawn_env(8)["OMP_NUM_THREADS"] == "3"


def test_free_ports_differ():
    bench = _bench()
    assert 1024 < bench.free_port() < 65536


CHILD = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import bench                                  # records the start-up affinity
from threadpoolctl import threadpool_info
import numpy as np

def pools():
    return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])

start = sorted(os.sched_getaffinity(0))
os.sched_setaffinity(0, start[:1])            # what bind_rank_to_numa_node does, taken to the extreme
before = (pools(), len(os.sched_getaffinity(0)))
with bench.host_cores():
    inside = (pools(), len(os.sched_getaffinity(0)), bench.host_threads())
    np.linalg.eigh(np.eye(64))                # the pools are usable at that size
after = (pools(), len(os.sched_getaffinity(0)))
print(json.dumps({{"start": len(start), "before": before, "inside": inside, "after": after}}))
"""


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="needs sched_setaffinity")
def test_cpu_legs_get_the_start_up_cores_back():
    if len(os.sched_getaffinity(0)) < 2:
        pytest.skip("one CPU only")
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")   # as exported by torch.distributed.run
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["before"] == [1, 1]                                 # one BLAS thread on one core: the broken leg
    assert out["inside"] == [out["start"], out["start"], out["start"]]
    assert out["after"] == [1, 1]                                  # the GPU-driving phase gets its binding back

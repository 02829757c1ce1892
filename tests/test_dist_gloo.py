"""Distributed plumbing tier on CPU/gloo, world_size=2 (BASELINE.json config 1):
real ``torch.distributed.run`` launches of tests/dist_worker.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")
_port = [29600 + (os.getpid() % 200)]


def _launch(scenario, out, nproc=2, timeout=600, extra_env=None):
    _port[0] += 1
    env = dict(os.environ)
    env.update(OMP_NUM_THREADS="2", DTP_PG_TIMEOUT_S="60")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
           "--master-addr", "127.0.0.1", "--master-port", str(_port[0]), WORKER, scenario, str(out)]
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_reducer_equals_large_batch_gloo(tmp_path, world):
    """world 4 rehearses the rank counts of the 4-/8-GPU runs the driver launches (same bucket
    planning, broadcast, averaging and no_sync logic; only the transport differs on GPUs)."""
    r = _launch("reducer", tmp_path, nproc=world)
    assert r.returncode == 0, r.stdout[-4000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == ["ok.%d" % i for i in range(world)]


def test_trainer_vgg16_synthetic_gloo_world2(tmp_path):
    r = _launch("trainer", tmp_path, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == ["ok.0", "ok.1"]


def test_killed_rank_fails_fast_instead_of_hanging(tmp_path):
    r = _launch("fault", tmp_path, timeout=600, extra_env=dict(DTP_PG_TIMEOUT_S="20"))
    assert r.returncode != 0                                  # surfaced as an error ...
    assert not os.path.exists(os.path.join(tmp_path, "ok.0"))  # ... and nobody reported success

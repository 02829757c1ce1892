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


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_is_bit_identical_to_allreduce_gloo(tmp_path, world):
    r = _launch("sharded", tmp_path, nproc=world)
    assert r.returncode == 0, r.stdout[-4000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == ["ok.%d" % i for i in range(world)]


def test_validation_with_fewer_samples_than_ranks_gloo(tmp_path):
    r = _launch("small_val", tmp_path, nproc=2, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == ["ok.0", "ok.1"]


def test_survivor_of_a_killed_rank_raises_within_the_timeout(tmp_path):
    """No elastic agent, no launcher: two bare worker processes.  Rank 1 kills itself
    (``os._exit``) at global step 1; NOTHING terminates rank 0 from outside — it must notice the
    dead peer on its own, raise within ``DTP_PG_TIMEOUT_S`` and exit non-zero with a message that
    names the communication failure."""
    import time
    port = _port[0] = _port[0] + 1
    timeout_s = 20
    procs, logs = [], []
    for rank in range(2):
        env = dict(os.environ, OMP_NUM_THREADS="2", DTP_PG_TIMEOUT_S=str(timeout_s), RANK=str(rank),
                   LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        log = open(os.path.join(tmp_path, "rank%d.log" % rank), "w")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, "-u", WORKER, "fault", str(tmp_path)], cwd=ROOT, env=env,
                                      stdout=log, stderr=subprocess.STDOUT))
    assert procs[1].wait(timeout=600) == 17                   # the injected fault
    t_dead = time.time()
    try:
        rc0 = procs[0].wait(timeout=timeout_s + 90)           # generous slack for a loaded CI box
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for log in logs:
            log.close()
    elapsed = time.time() - t_dead
    out0 = open(os.path.join(tmp_path, "rank0.log")).read()
    assert rc0 != 0, out0[-3000:]                             # surfaced as an error, not a hang ...
    assert elapsed < timeout_s + 90
    assert not os.path.exists(os.path.join(tmp_path, "ok.0"))  # ... nobody reported success ...
    low = out0.lower()                                         # ... and the error names the cause
    assert any(k in low for k in ("timed out", "timeout", "connection closed", "connection reset", "peer",
                                  "broken pipe")), out0[-3000:]

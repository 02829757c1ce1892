"""Real multi-GPU tier: one process per GPU, RCCL process group, native RCCL gradient transport.
Runs only where at least two GPUs are visible (the development boxes of this project have one, so
these are for the multi-GPU node: `python -m pytest tests/test_multi_gpu.py -m gpu`).  World sizes 2
and — when the box has them — 4 and 8."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")
NGPU = torch.cuda.device_count()          # does not initialise the GPU in this process
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NGPU < 2, reason="needs >= 2 GPUs (found %d)" % NGPU)]
WORLDS = [w for w in (2, 4, 8) if w <= NGPU]
_port = [29900 + (os.getpid() % 50)]


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DTP_PG_TIMEOUT_S="300")
    env.update(kw)
    return env


@pytest.mark.parametrize("world", WORLDS or [2])
def test_fused_model_data_parallel_over_native_rccl(tmp_path, world):
    """The scenario the one-GPU box rehearses over gloo with a shared device (gpu_shared), here with
    a GPU per rank: parameter broadcast, arena-range buckets, kernel-side ready notifications, the
    C++ bucket scheduler issuing ncclAllReduce on the comm stream while backward continues; the
    averaged gradients and the parameters must agree on every rank after every step."""
    _port[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(_port[0]), WORKER, "gpu_shared", str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(DTP_TEST_MULTI_GPU="1", DTP_TRANSPORT="native"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-6000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == sorted("ok.%d" % i for i in range(world))
    losses = [json.load(open(os.path.join(tmp_path, "losses.%d" % i)))["losses"] for i in range(world)]
    assert all(l[-1] < l[0] for l in losses), losses


@pytest.mark.parametrize("world", WORLDS or [2])
def test_every_sync_mode_agrees_across_ranks_and_with_allreduce(tmp_path, world):
    """VGG16-BN through all-reduce, per-bucket optimizer, sharded optimizer and the hipGraph-captured
    data-parallel step on the native transport (tests/dist_worker.py gpu_modes): cross-rank equality of
    everything the kernels read after every step, per-bucket and hipGraph == all-reduce bit for bit."""
    _port[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(_port[0]), WORKER, "gpu_modes", str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(DTP_TEST_MULTI_GPU="1", DTP_TRANSPORT="native"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-6000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == sorted("ok.%d" % i for i in range(world))


def test_main_hip_graph_data_parallel_two_gpus(tmp_path):
    """`main.py --nproc 2 --hip-graph`: the Trainer's captured step INCLUDING the RCCL collectives."""
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--nproc", "2", "--synthetic", "--synthetic-size", "256",
           "--epochs", "2", "--batch-size", "32", "--image-size", "32", "--no-validate", "--no-progress", "--hip-graph",
           "--save-folder", str(tmp_path / "runs")]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(DTP_TRANSPORT="native"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-6000:]
    assert "Finished!" in r.stdout and os.path.exists(tmp_path / "runs" / "weights" / "checkpoint_epoch_2.pth")


@pytest.mark.parametrize("extra", [[], ["--no-bucket-optimizer"], ["--shard-optimizer"], ["--grad-accum", "2"]])
def test_bench_self_launch_two_gpus_native_transport(extra):
    """`python bench.py --gpus 2` from a bare shell: the self-launcher starts one fresh worker per
    GPU; the JSON line must report both ranks, the native transport (no fall-back) and a passed
    transport self-check."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--batch-per-gpu", "32"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=_env(DTP_TRANSPORT="native"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-6000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["grad_transport_selfcheck_ok"] is True
    assert d["grad_transport"] == "NativeRcclTransport" and d["native_transport_fell_back"] is False
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]

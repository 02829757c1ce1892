"""The self-launcher (distributed_training_pytorch_amd/launch.py) and the entry points built on it:
``python bench.py --gpus N`` / ``python main.py --nproc N`` from a bare shell (reference run.sh:9-14
does this with torchrun).  CPU/gloo; no elastic agent anywhere."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENV = dict(os.environ, OMP_NUM_THREADS="2", DTP_PG_TIMEOUT_S="120")
ENV.pop("WORLD_SIZE", None)
ENV.pop("RANK", None)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_fresh_workers_cpu():
    """Plain ``python bench.py --gpus 2``: the parent spawns two workers, rank 0 prints ONE JSON line
    that proves both ranks took part; the sharded-optimizer variant reaches the same loss."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--batch-per-gpu", "4",
            "--image-size", "32", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(base, cwd=ROOT, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["grad_transport_selfcheck_ok"] is True
    assert j["config"]["global_batch"] == 8 and j["config"]["parallelism"] == "dp2"
    assert j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0 and j["final_loss"] == j["final_loss"]
    assert len(j["bucket_mbytes"]) >= 3 and abs(sum(j["bucket_mbytes"]) - 134272835 * 4 / 2 ** 20) < 2.0
    assert j["native_transport_fell_back"] is False and j["grad_transport"] == "TorchPGTransport"
    r2 = subprocess.run(base + ["--shard-optimizer"], cwd=ROOT, env=ENV, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    j2 = _json_line(r2.stdout)
    assert "sharded per bucket" in j2["config"]["optimizer"]
    assert j2["final_loss"] == j["final_loss"]                  # world 2: the same sums in the same order


DRIVER_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
               "vs_baseline", "dtype", "data", "config")


def test_bench_under_torch_distributed_run_cpu():
    """The driver's N > 1 launch: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`` — ranks come from
    the environment, rank 0 prints the result, and stdout carries NOTHING else (gloo's connection
    notices and RCCL's banner go to stderr)."""
    from distributed_training_pytorch_amd.launch import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu",
           "--batch-per-gpu", "4", "--image-size", "32", "--steps", "2", "--warmup", "1"]
    # (DTP_PREFLIGHT=force: the pre-flight rehearsal that GPU runs do at N > 1 runs here too — a throw-away
    # child per rank on another rendezvous port before the real run — and must leave stdout untouched)
    r = subprocess.run(cmd, cwd=ROOT, env=dict(ENV, DTP_PREFLIGHT="force"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-3000:]                    # ONE line on stdout, the JSON
    j = json.loads(lines[0])
    assert all(k in j for k in DRIVER_KEYS), sorted(j)
    assert j["preflight"]["ok"] is True and j["preflight"]["all_ranks_ok"] is True
    assert j["native_transport_fell_back"] is False
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["config"]["global_batch"] == 8
    assert j["grad_transport_selfcheck_ok"] is True


def _torchrun_bench(extra_env, port=None):
    from distributed_training_pytorch_amd.launch import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port or free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--device", "cpu", "--batch-per-gpu", "4", "--image-size", "32", "--steps", "2", "--warmup", "1"]
    return subprocess.run(cmd, cwd=ROOT, env=dict(ENV, **extra_env), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=900)


def test_bench_preflight_rehearsal_under_torchrun_cpu():
    """N > 1 under a foreign launcher: every rank rehearses the run in a throw-away child on another
    rendezvous port (bench.preflight_native / launch.preflight) before the real one.  (The passing rehearsal is
    part of test_bench_under_torch_distributed_run_cpu.)  Here it (a) fails on every rank, (b) hangs on every
    rank and is killed at its deadline: the real run still happens, on the process-group transport, and its
    JSON says why.  CPU / gloo stands in for the GPUs (DTP_PREFLIGHT=force)."""
    for inject, timeout_s, word in (("fail", "120", "exited with 3"), ("hang", "4", "killed")):
        r = _torchrun_bench(dict(DTP_PREFLIGHT="force", DTP_PREFLIGHT_INJECT=inject, DTP_PREFLIGHT_TIMEOUT_S=timeout_s))
        assert r.returncode == 0, r.stderr[-3000:]
        j = _json_line(r.stdout)
        assert j["preflight"]["ok"] is False and j["preflight"]["all_ranks_ok"] is False
        assert j["native_transport_fell_back"] is True and word in j["native_transport_fallback_reason"]
        assert j["grad_transport"] == "TorchPGTransport" and j["ranks_seen"] == 2 and j["grad_transport_selfcheck_ok"]
        assert "pre-flight rehearsal" in r.stderr


def test_preflight_port_is_a_pure_function_of_the_job_port():
    from distributed_training_pytorch_amd.launch import preflight_port
    for p in (29500, 12345, 65535, 1024):
        q = preflight_port(p)
        assert q == preflight_port(str(p)) and 20000 <= q < 40000 and q != p


def test_bench_single_process_path_is_unchanged_cpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--batch-per-gpu", "2",
                        "--image-size", "32", "--steps", "1", "--warmup", "1"], cwd=ROOT, env=ENV,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1 and all(k in j for k in DRIVER_KEYS)
    assert j["n_gpus"] == 1 and j["ranks_seen"] == 1 and j["config"]["parallelism"] == "dp1"
    # WORLD_SIZE says 1 but --gpus says 2: refuse instead of benchmarking the wrong job
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu"], cwd=ROOT,
                         env=dict(ENV, WORLD_SIZE="1", RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert bad.returncode != 0 and "does not match WORLD_SIZE" in bad.stderr


def _write(tmp_path, name, body):
    p = os.path.join(str(tmp_path), name)
    with open(p, "w") as f:
        f.write(textwrap.dedent(body))
    return p


def test_launcher_stops_the_job_when_a_worker_fails(tmp_path):
    """One worker exits non-zero: the others (which would run for minutes) are terminated at once
    and the launcher reports that worker's status."""
    from distributed_training_pytorch_amd import launch
    script = _write(tmp_path, "w.py", """
        import os, sys, time
        rank = int(os.environ["RANK"])
        assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "NCCL_MIN_NCHANNELS" in os.environ
        open(os.path.join(sys.argv[1], "started.%d" % rank), "w").close()
        if rank == 1:
            time.sleep(0.5)
            sys.exit(9)
        time.sleep(300)
    """)
    t0 = time.time()
    rc = launch.spawn_workers(script, [str(tmp_path)], 3, log=lambda m: None)
    assert rc == 9 and time.time() - t0 < 60
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("started.")) == ["started.0", "started.1", "started.2"]


def test_launcher_falls_back_to_torch_transport_and_restarts(tmp_path):
    from distributed_training_pytorch_amd import launch
    # (a) a worker gives up on the native transport: ONE relaunch of everybody with DTP_TRANSPORT=torch
    script = _write(tmp_path, "fb.py", """
        import os, sys
        from distributed_training_pytorch_amd.launch import EXIT_NATIVE_BRINGUP
        rank, mode = os.environ["RANK"], os.environ.get("DTP_TRANSPORT", "auto")
        open(os.path.join(sys.argv[1], "run.%s.%s" % (mode, rank)), "w").close()
        if mode != "torch":
            if rank == "1":          # give up only once rank 0 is seen running (the launcher stops it then)
                import time
                t0 = time.time()
                while not os.path.exists(os.path.join(sys.argv[1], "run.%s.0" % mode)) and time.time() - t0 < 60:
                    time.sleep(0.05)
            sys.exit(EXIT_NATIVE_BRINGUP if rank == "1" else 0)
        assert os.environ["DTP_NATIVE_FELL_BACK"] == "1"
    """)
    env = dict(PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    msgs = []
    assert launch.spawn_workers(script, [str(tmp_path)], 2, extra_env=env, log=msgs.append) == 0
    runs = sorted(f for f in os.listdir(tmp_path) if f.startswith("run."))
    assert runs == ["run.auto.0", "run.auto.1", "run.torch.0", "run.torch.1"], runs
    assert any("DTP_TRANSPORT=torch" in m for m in msgs)
    # forced native: no silent fallback
    assert launch.spawn_workers(script, [str(tmp_path)], 2, extra_env=dict(env, DTP_TRANSPORT="native"),
                                log=msgs.append) == launch.EXIT_NATIVE_BRINGUP
    # (b) max_restarts: a failing job is re-run with DTP_RESTART_COUNT incremented
    script2 = _write(tmp_path, "rs.py", """
        import os, sys
        n = int(os.environ["DTP_RESTART_COUNT"])
        open(os.path.join(sys.argv[1], "attempt.%d.%s" % (n, os.environ["RANK"])), "w").close()
        sys.exit(0 if n >= 2 else 3)
    """)
    assert launch.spawn_workers(script2, [str(tmp_path)], 2, max_restarts=1, log=msgs.append) == 3
    assert launch.spawn_workers(script2, [str(tmp_path)], 2, max_restarts=2, log=msgs.append) == 0


def test_main_self_launch_restart_resumes_from_last_snapshot(tmp_path):
    """``python main.py --nproc 2 --max-restarts 1``: rank 1 dies in the second epoch; the
    launcher re-runs the job and the restarted workers resume from runs/weights/last.pth on their
    own — the run still ends with 'Finished!' and the last epoch's snapshot."""
    save = str(tmp_path / "runs")
    env = dict(ENV, FAULT_INJECT_STEP="3", FAULT_INJECT_RANK="1", DTP_PG_TIMEOUT_S="30")
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--nproc", "2", "--max-restarts", "1", "--backend", "gloo",
           "--synthetic", "--synthetic-size", "16", "--image-size", "32", "--epochs", "3", "--batch-size", "8",
           "--lr", "0.01", "--save-period", "1", "--save-folder", save, "--no-progress"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "restart 1 of 1" in r.stdout
    log = open(os.path.join(save, "logfile.log")).read()
    assert "Resumed from" in log and "last.pth at epoch 1" in log and "Finished!" in log
    from distributed_training_pytorch_amd.utils.checkpoint import load_snapshot
    assert load_snapshot(os.path.join(save, "weights", "last.pth"))["epoch"] == 3

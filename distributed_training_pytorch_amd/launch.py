"""Self-contained multi-process launcher: one fresh worker process per GPU, no elastic agent.

Parity target: the reference starts its ranks with ``torchrun --nproc_per_node=1 --nnodes=4
--node_rank=0 --master_addr=... --master_port=... main.py`` after exporting its NCCL variables
(``run.sh:1-14``), and reads ``LOCAL_RANK`` / ``RANK`` / ``WORLD_SIZE`` in the workers
(``trainer/trainer.py:48-50,76``).  This module provides the same contract from plain Python so
``python bench.py --gpus 8`` and ``python main.py --nproc 8`` work from a bare shell:

  * the parent NEVER touches the GPU (it does not even import torch): workers are brand-new
    processes started with ``subprocess`` — a process that has initialised HIP is never re-exec'ed
    or forked;
  * the RCCL / HIP environment defaults (``rccl_env_defaults``) are exported to the workers here,
    not only by ``run.sh``;
  * every worker's exit status is watched: the first non-zero exit terminates the remaining
    workers (exact PIDs, each worker is its own session) and becomes the launcher's exit status;
  * a worker that exits with ``EXIT_NATIVE_BRINGUP`` (its native RCCL transport did not come up
    before the deadline) triggers ONE relaunch of the whole job with ``DTP_TRANSPORT=torch``;
  * ``max_restarts`` re-runs a failed job (fresh rendezvous port, ``DTP_RESTART_COUNT`` exported)
    so ``main.py`` can resume from ``last.pth`` — elastic recovery without the elastic agent.
"""
import os
import signal
import socket
import subprocess
import sys
import time

EXIT_NATIVE_BRINGUP = 75      # a worker gave up on the native RCCL transport (see data_parallel.py)


def rccl_env_defaults(nnodes=1):
    """Environment the RCCL path wants on an MI355X node (xGMI mesh, dmabuf IPC).  Values are
    defaults: anything already set by the operator wins.  Also applied inside the workers by
    ``parallel.env.apply_rccl_env_defaults`` for launches through torchrun / run.sh."""
    env = {
        # dmabuf IPC: without it RCCL's cross-process buffer sharing fails on this pool
        "HSA_ENABLE_IPC_MODE_LEGACY": "0",
        # 7 xGMI links per GPU: a single ring uses one of them; ask for enough channels that the
        # rings/trees RCCL builds cover every link
        "NCCL_MIN_NCHANNELS": "32",
        # surface a dead peer as an error in the process group instead of a silent hang
        "TORCH_NCCL_ASYNC_ERROR_HANDLING": "1",
        "TORCH_FR_BUFFER_SIZE": "2000",              # flight recorder (reference run.sh:8 sets its older name)
        # NCCL_DEBUG stays unset here on purpose: RCCL writes its WARN lines (e.g. the iommu=pt notice
        # many hosts trigger) to STDOUT, and bench.py promises one machine-readable line there;
        # run.sh, the operator-facing launcher, exports NCCL_DEBUG=WARN
    }
    if nnodes > 1:
        env["NCCL_SOCKET_IFNAME"] = "eno1"            # reference run.sh:3 (bootstrap interface)
    return env


def free_port(host="127.0.0.1"):
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        s.bind((host, 0))
        return s.getsockname()[1]
    finally:
        s.close()


def _terminate(procs, grace_s=5.0):
    """Stop exactly the processes this launcher started (never by pattern)."""
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
    deadline = time.time() + grace_s
    for p in procs:
        while p.poll() is None and time.time() < deadline:
            time.sleep(0.05)
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()


def _run_once(cmd, nproc, env, master_addr, master_port, node_rank, nnodes, log=None):
    procs = []
    world = nproc * nnodes
    for lr in range(nproc):
        e = dict(env)
        e.update(RANK=str(node_rank * nproc + lr), LOCAL_RANK=str(lr), WORLD_SIZE=str(world),
                 LOCAL_WORLD_SIZE=str(nproc), MASTER_ADDR=master_addr, MASTER_PORT=str(master_port),
                 DTP_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen(cmd, env=e, start_new_session=True))

    def _on_signal(signum, frame):
        _terminate(procs)
        sys.exit(128 + signum)

    old = {s: signal.signal(s, _on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.05)
            for p in list(live):
                r = p.poll()
                if r is None:
                    continue
                live.remove(p)
                if r != 0 and rc == 0:
                    rc = r if r > 0 else 128 - r
                    if log:
                        log("worker pid %d (local rank %d) exited with %d: stopping the other %d workers"
                            % (p.pid, procs.index(p), r, len(live)))
                    _terminate(live)
                    live = []
                    break
    finally:
        _terminate(procs)
        for s, h in old.items():
            signal.signal(s, h)
    return rc


def spawn_workers(script, argv, nproc, nnodes=1, node_rank=0, master_addr="127.0.0.1", master_port=None,
                  max_restarts=0, extra_env=None, python=None, log=None):
    """Run ``python script *argv`` as ``nproc`` local ranks; returns the job's exit status (0 = all
    workers succeeded).  Must be called before the calling process touches the GPU."""
    if log is None:
        def log(msg):
            print("[dtp-launch] %s" % msg, file=sys.stderr, flush=True)
    env = dict(os.environ)
    for k, v in rccl_env_defaults(nnodes).items():
        env.setdefault(k, v)
    env.update(extra_env or {})
    cmd = [python or sys.executable, "-u", script] + list(argv)
    fixed_port = master_port is not None
    attempt = restarts = 0
    fell_back = False
    while True:
        port = int(master_port) if fixed_port else free_port(master_addr if nnodes == 1 else "0.0.0.0")
        e = dict(env, DTP_RESTART_COUNT=str(restarts))
        rc = _run_once(cmd, nproc, e, master_addr, port, node_rank, nnodes, log)
        attempt += 1
        if rc == 0:
            return 0
        if rc == EXIT_NATIVE_BRINGUP and not fell_back and env.get("DTP_TRANSPORT", "auto").lower() != "native":
            log("native RCCL transport did not come up: relaunching %d fresh workers with DTP_TRANSPORT=torch" % nproc)
            env["DTP_TRANSPORT"] = "torch"
            env["DTP_NATIVE_FELL_BACK"] = "1"
            fell_back = True
            continue
        if restarts < max_restarts:
            restarts += 1
            log("job failed with %d: restart %d of %d" % (rc, restarts, max_restarts))
            continue
        return rc


def preflight_port(master_port):
    """Rendezvous port of the pre-flight children: a fixed function of the job's port, so that every
    rank derives the same one without talking to the others."""
    p = int(master_port)
    return 20000 + (p * 7 + 4099) % 20000


def preflight(script, argv, timeout_s=240.0, extra_env=None, python=None):
    """Rehearse a multi-rank job in ONE throw-away child process per rank before the real run.

    Called by every rank of a job (``RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT`` in the
    environment) BEFORE it touches the GPU: starts ``python script *argv`` with the same rank
    variables on another rendezvous port and waits for it.  The children form a job of their own, so a
    code path that cannot be exercised on this rank's own resources alone (a second RCCL communicator
    between exactly these GPUs, the C++ bucket scheduler on it) either works end to end there, or
    fails / hangs THERE — in processes that are killed (exact PIDs) without harming the caller, which
    then picks the conservative path.  A foreign launcher (torchrun) restarts nothing, so this is the
    only place where such a failure can still cost a fallback instead of the job.

    Returns ``(ok, reason)``; never raises for a failing child."""
    import tempfile
    env = dict(os.environ)
    for k, v in rccl_env_defaults().items():
        env.setdefault(k, v)
    env.update(extra_env or {})
    env["MASTER_PORT"] = str(preflight_port(os.environ.get("MASTER_PORT", "29500")))
    # the children rendezvous among themselves: rank 0's child hosts the store, not the elastic agent
    for k in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"):
        env.pop(k, None)
    env["DTP_PREFLIGHT"] = "0"
    cmd = [python or sys.executable, "-u", script] + list(argv)
    t0 = time.time()
    with tempfile.TemporaryFile(mode="w+b") as err:
        try:
            proc = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=err, start_new_session=True)
        except OSError as e:
            return False, "pre-flight child could not be started: %s" % e
        try:
            rc = proc.wait(timeout=timeout_s)
            reason = None if rc == 0 else "pre-flight child exited with %d after %.0f s" % (rc, time.time() - t0)
        except subprocess.TimeoutExpired:
            _terminate([proc], grace_s=3.0)
            rc, reason = -1, "pre-flight child still running after %.0f s: killed" % timeout_s
        if rc != 0:
            try:
                err.seek(0)
                tail = err.read()[-1500:].decode("utf-8", "replace").strip().splitlines()
                if tail:
                    reason += " (%s)" % " | ".join(tail[-3:])
            except Exception:
                pass
    return rc == 0, reason

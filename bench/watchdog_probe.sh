# experiment: two ranks share cuda:0, transport=auto -> native bring-up cannot work (duplicate GPU);
# the watchdog must degrade to the process-group transport within the deadline.
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/wd && rm -f /tmp/wd/*
cd $R
DTP_TRANSPORT=auto DTP_NATIVE_INIT_TIMEOUT_S=15 DTP_PG_TIMEOUT_S=60 OMP_NUM_THREADS=2 timeout -k 10 150 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29811 \
  tests/dist_worker.py gpu_shared /tmp/wd > gpurun_out/watchdog.log 2>&1
echo "exit $?"; ls /tmp/wd

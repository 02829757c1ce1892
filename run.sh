#!/usr/bin/env bash
# torchrun launcher (reference run.sh:1-14), re-tuned for one MI355X node: 8 GPUs on a
# fully-connected xGMI mesh, RCCL as the "nccl" backend.  The reference's settings target
# 4 hosts x 1 GPU over TCP (NCCL_IB_DISABLE=1, NCCL_ALGO=Ring, NCCL_PROTO=Simple): a single
# ring would use 1 of the 7 xGMI links per GPU, so algorithm/protocol are left to RCCL here
# and only the channel floor is raised.
#
#   NGPU=8 bash run.sh --synthetic --batch-size 2048
#   NNODES=4 NODE_RANK=2 MASTER_ADDR=10.0.0.1 bash run.sh        # multi-node, one line per node
set -euo pipefail

NGPU="${NGPU:-8}"
NNODES="${NNODES:-1}"
NODE_RANK="${NODE_RANK:-0}"
MASTER_ADDR="${MASTER_ADDR:-127.0.0.1}"
MASTER_PORT="${MASTER_PORT:-29500}"
MAX_RESTARTS="${MAX_RESTARTS:-0}"

export NCCL_DEBUG="${NCCL_DEBUG:-WARN}"                       # INFO prints topology / channel graph
export NCCL_MIN_NCHANNELS="${NCCL_MIN_NCHANNELS:-32}"         # spread rings over all xGMI links
export HSA_ENABLE_IPC_MODE_LEGACY="${HSA_ENABLE_IPC_MODE_LEGACY:-0}"   # dmabuf IPC (required on this pool)
export TORCH_NCCL_ASYNC_ERROR_HANDLING="${TORCH_NCCL_ASYNC_ERROR_HANDLING:-1}"
export TORCH_FR_BUFFER_SIZE="${TORCH_FR_BUFFER_SIZE:-2000}"                 # flight recorder entries (reference run.sh:8)
export DTP_PG_TIMEOUT_S="${DTP_PG_TIMEOUT_S:-600}"
if [ "${NNODES}" -gt 1 ]; then
  export NCCL_SOCKET_IFNAME="${NCCL_SOCKET_IFNAME:-eno1}"     # bootstrap / inter-node interface
fi

# Auto-resume is decided INSIDE each (re)started worker: main.py picks up
# <save-folder>/weights/last.pth when TORCHELASTIC_RESTART_COUNT > 0 (see main.py: pick_snapshot).

exec python -m torch.distributed.run \
  --nproc_per_node="${NGPU}" \
  --nnodes="${NNODES}" \
  --node_rank="${NODE_RANK}" \
  --master_addr="${MASTER_ADDR}" \
  --master_port="${MASTER_PORT}" \
  --max_restarts="${MAX_RESTARTS}" \
  main.py "$@"

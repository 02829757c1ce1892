"""Fused flat SGD alone on the GPU over a VGG16-sized arena (134 M parameters, bf16 gradients).
(What a full-chip optimizer grid does to a kernel launched next to it is in profiles/76_*: a replay timeline.)
DTP_OPT_GRID_CAP selects the optimizer grid (512 = 2 workgroups per CU, the default; 4096 = the old grid)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd import _lib  # noqa: E402


def main():
    n = 134_272_832 // 256 * 256
    dev = torch.device("cuda")
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev).to(torch.bfloat16)
    m = torch.zeros(n, device=dev)
    lp = torch.empty(n, device=dev, dtype=torch.bfloat16)
    hip = _lib.hip()
    vp = _lib.c_void_p

    def sgd(stream):
        rc = hip.dtp_sgd_flat(vp(p.data_ptr()), vp(g.data_ptr()), 1, vp(m.data_ptr()), vp(lp.data_ptr()), n,
                              0.001, 0.9, 0.0, 1e-4, 1.0, 0, 0, vp(stream.cuda_stream))
        assert rc == 0
    s0 = torch.cuda.current_stream()
    for _ in range(3):
        sgd(s0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        sgd(s0)
    e1.record()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) / 20
    gb = n * (4 + 2 + 4 + 4 + 4 + 2) / 1e9
    print(json.dumps({"grid_cap": os.environ.get("DTP_OPT_GRID_CAP", "512 (default)"), "sgd_ms": round(alone, 4),
                      "TB_per_s": round(gb / alone, 3)}))


if __name__ == "__main__":
    main()

"""Board power and shader clock (rocm-smi samples) while ONE kernel runs back to back for a few seconds:
is a kernel power-limited?  python bench/power_probe.py    (one JSON line per kernel)"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as K  # noqa: E402

BF16 = torch.bfloat16
N = int(os.environ.get("N", "256"))
SECONDS = float(os.environ.get("SECONDS", "4"))


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        p = [float(v) for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()]
        s = [v for k, v in card.items() if k.lower().startswith("sclk clock speed")]
        mhz = float(s[0].strip("()").lower().replace("mhz", "")) if s else None
        return (p[0] if p else None), mhz
    except Exception as e:          # noqa: BLE001
        return None, None


def run(name, fn, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.3)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    pw = [p for p, _ in samples[2:] if p is not None]
    ck = [c for _, c in samples[2:] if c is not None]
    print(json.dumps(dict(kernel=name, batch=N, tflops=round(flops * n / dt / 1e12, 1) if flops else None,
                          ms=round(1e3 * dt / n, 4), power_w_mean=round(sum(pw) / len(pw), 1) if pw else None,
                          power_w_max=max(pw) if pw else None, sclk_mhz_mean=round(sum(ck) / len(ck)) if ck else None,
                          samples=len(pw))), flush=True)


def conv(C, Co, HW):
    x = torch.randn(N, HW, HW, C, device="cuda").to(BF16)
    w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
    b = torch.zeros(Co, device="cuda")
    return (lambda: K.conv3x3_fprop(x, w, b, True)), 2.0 * N * HW * HW * C * Co * 9


print(json.dumps(dict(idle=smi())), flush=True)
for name, (C, Co, HW) in [("conv3_2 fprop igemm<256,256>", (256, 256, 56)), ("conv4_2 fprop igemm<256,256>", (512, 512, 28)),
                          ("conv2_2 fprop igemm<512,128>", (128, 128, 112)), ("conv1_2 fprop+pool conv_c64<64>", (64, 64, 224))]:
    fn, fl = conv(C, Co, HW)
    if C == 64:
        x = torch.randn(N, HW, HW, C, device="cuda").to(BF16)
        w = (torch.randn(Co, 9 * C, device="cuda") * 0.04).to(BF16)
        b = torch.zeros(Co, device="cuda")
        fn = lambda: K.conv3x3_relu_pool_fprop(x, w, b)       # noqa: E731
    run(name, fn, fl)
    torch.cuda.empty_cache()
px = torch.randn(N, 112, 112, 128, device="cuda").to(BF16)
py, idx = K.maxpool2x2_fwd(px, want_idx=True)
pdy = torch.randn_like(py)
K.fuse_colsum = False
run("maxpool2x2 backward (memory-bound)", lambda: K.maxpool2x2_bwd(None, pdy, relu_mask=True, idx=idx, in_shape=(N, 112, 112, 128)), 0)

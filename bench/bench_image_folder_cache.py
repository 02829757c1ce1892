"""Image-folder training from the HBM cache with device-side augmentation: throughput and GPU-side
input starvation of the real Trainer path (`ExampleTrainer`, uint8 input, `--device-cache`).

No dataset can be downloaded here, so the folder is written first: `--classes` x `--per-class` random
PNG files (the decode + resize cost is paid ONCE, while the cache is built; what is measured is the
steady state from the second epoch on).  Prints one JSON line per epoch (from the Trainer's metrics
file) and a summary line."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_folder(root, labels, per_class, size):
    from PIL import Image
    rs = np.random.RandomState(0)
    base = rs.randint(0, 256, size=(len(labels), size, size, 3)).astype(np.int16)
    for c, lb in enumerate(labels):
        os.makedirs(os.path.join(root, lb), exist_ok=True)
        for i in range(per_class):
            img = np.clip(base[c] + rs.randint(-40, 40, size=(size, size, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, lb, "%05d.png" % i), compress_level=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-class", type=int, default=1024)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--out", default="gpurun_out/image_folder_cache.jsonl")
    args = ap.parse_args()
    from distributed_training_pytorch_amd.trainer.example_trainer import ExampleTrainer
    labels = ["c%d" % i for i in range(args.classes)]
    tmp = tempfile.mkdtemp(prefix="dtp_imgs_")
    t0 = time.time()
    write_folder(os.path.join(tmp, "train"), labels, args.per_class, args.size)
    t_write = time.time() - t0
    metrics = os.path.join(tmp, "metrics.jsonl")
    tr = ExampleTrainer(train_path=os.path.join(tmp, "train"), val_path=os.path.join(tmp, "train"), labels=labels,
                        height=args.size, width=args.size, max_epoch=args.epochs, batch_size=args.batch, pin_memory=True,
                        have_validate=False, save_period=1000, save_folder=os.path.join(tmp, "runs"), progress=False,
                        lr=0.001, uint8_input=True, cache_dataset_on_device=True, grad_dtype=torch.bfloat16,
                        metrics_file=metrics)
    from distributed_training_pytorch_amd.data.device_cache import DeviceCachedLoader
    assert isinstance(tr.train_dataloader, DeviceCachedLoader) and tr.train_dataloader.augmenter is not None
    t0 = time.time()
    tr.train()
    torch.cuda.synchronize()
    recs = [json.loads(l) for l in open(metrics)]
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        for r in recs:
            f.write(json.dumps(r) + "\n")
        steady = recs[1:] or recs
        summary = dict(summary=True, images=args.classes * args.per_class, batch=args.batch, image_size=args.size,
                       write_folder_s=round(t_write, 1), train_s=round(time.time() - t0, 1),
                       steady_images_per_sec=round(float(np.mean([r["images_per_sec"] for r in steady])), 1),
                       steady_data_starved_ms_per_step=round(float(np.mean([r["data_starved_ms_per_step"] for r in steady])), 4),
                       steady_data_wait_ms_per_step=round(float(np.mean([r["data_wait_ms_per_step"] for r in steady])), 3),
                       augmentation="device: rot90 / h-flip / v-flip / brightness-contrast / gamma per (epoch, sample)")
        f.write(json.dumps(summary) + "\n")
    print(json.dumps(summary))


if __name__ == "__main__":
    main()

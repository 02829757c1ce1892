"""Offline evaluation entry point (reference eval.py:1-72): load ``runs/weights/last.pth``'s
``model_state_dict`` into VGG16, run every image under ``data/test/<label>/``, print
``EVALUATION ACCURACY RESULTS: TOP-1=..% --- TOP-2=..%``.

Differences: batched inference under ``torch.no_grad``; ragged class
folders are fine; top-k is computed from logits on the device (k = min(2, n_classes)); no
cv2 / torchvision / sklearn dependency; ``--synthetic`` evaluates on the deterministic synthetic
test split when no image folder exists (no dataset can be downloaded here).
"""
import argparse
import os

import torch

from distributed_training_pytorch_amd.data.image_folder import ExampleDataset
from distributed_training_pytorch_amd.data.synthetic import SyntheticImageDataset
from distributed_training_pytorch_amd.utils.checkpoint import load_snapshot
from model.vgg16 import VGG16

DATA_FOLDER = "data/test"
MODEL_PATH = "runs/weights/last.pth"
DEVICE = "cuda:0" if torch.cuda.is_available() else "cpu"
LABEL_LIST = ["cat", "dog", "snake"]


def topk_hits(logits, labels, k):
    topk = logits.float().topk(k, dim=-1).indices
    return (topk == labels.view(-1, 1)).any(dim=-1).sum().item()


def evaluate(model, dataset, device, batch_size=64, k=2):
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False)
    n = top1 = topk = 0
    model.eval()
    with torch.no_grad():
        for img, lb in loader:
            img, lb = img.to(device), lb.to(device)
            out = model(img)
            top1 += topk_hits(out, lb, 1)
            topk += topk_hits(out, lb, k)
            n += lb.numel()
    return top1 / max(n, 1), topk / max(n, 1), n


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=DATA_FOLDER)
    ap.add_argument("--model", default=MODEL_PATH)
    ap.add_argument("--labels", default=",".join(LABEL_LIST))
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--device", default=DEVICE)
    ap.add_argument("--batch-norm", action="store_true")
    ap.add_argument("--fp8-classifier", action="store_true")
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate on N synthetic test samples")
    args = ap.parse_args(argv)
    labels = args.labels.split(",")
    device = torch.device(args.device)

    model = VGG16(3, len(labels), True, batch_norm=args.batch_norm, fp8_classifier=args.fp8_classifier)
    ckpt = load_snapshot(args.model)
    model.load_state_dict(ckpt["model_state_dict"])
    model = model.to(device)

    if args.synthetic or not os.path.isdir(args.data):
        n = args.synthetic or 256
        dataset = SyntheticImageDataset(n, len(labels), args.image_size, args.image_size, seed=2)
    else:
        dataset = ExampleDataset(args.data, labels, args.image_size, args.image_size, phase="val")
    k = min(2, len(labels))
    acc1, acck, n = evaluate(model, dataset, device, args.batch_size, k)
    print("EVALUATION ACCURACY RESULTS: TOP-1=%s%% --- TOP-%d=%s%%" % (acc1 * 100, k, acck * 100))
    return acc1, acck, n


if __name__ == "__main__":
    main()

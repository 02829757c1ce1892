"""Offline evaluation entry point (reference eval.py:1-72): load ``runs/weights/last.pth``'s
``model_state_dict`` into VGG16, run every image under ``data/test/<label>/``, print
``EVALUATION ACCURACY RESULTS: TOP-1=..% --- TOP-2=..%``.

Differences: batched inference under ``torch.no_grad``; ragged class
folders are fine; no cv2 / torchvision / sklearn dependency; ``--synthetic`` evaluates on the
deterministic synthetic test split when no image folder exists (no dataset can be downloaded here).

MI355X-first input path (the reference moves one image at a time with a blocking ``.to(DEVICE)`` and
reads the scores back per image, ``eval.py:55-64``): batches travel through the pinned-ring
``DevicePrefetcher`` on a copy stream — as uint8 pixels with ``--uint8-input``, normalised inside the
first convolution's ingest kernel — the hit counters stay on the device, and the host reads them back
ONCE at the end.

The metric is ``sklearn.metrics.top_k_accuracy_score``'s (reference ``eval.py:69-70``) including its
tie rule — among equal scores the HIGHER class index ranks first (stable ascending argsort, reversed)
— pinned by ``tests/test_unit_cpu.py::test_topk_hits_matches_sklearn_top_k_accuracy_score``.
"""
import argparse
import os

import torch

from distributed_training_pytorch_amd.data.image_folder import ExampleDataset
from distributed_training_pytorch_amd.data.prefetch import DevicePrefetcher
from distributed_training_pytorch_amd.data.synthetic import IMAGENET_MEAN, IMAGENET_STD, SyntheticImageDataset
from distributed_training_pytorch_amd.utils.checkpoint import load_snapshot
from model.vgg16 import VGG16

DATA_FOLDER = "data/test"
MODEL_PATH = "runs/weights/last.pth"
DEVICE = "cuda:0" if torch.cuda.is_available() else "cpu"
LABEL_LIST = ["cat", "dog", "snake"]


def topk_hits_tensor(logits, labels, k):
    """Number of rows whose true class is among the k best scores, as a 0-d int64 tensor on the
    scores' device (no host synchronisation).  Rank of the true class = classes that score higher +
    classes that score the same and have a higher index (sklearn's tie order)."""
    s = logits.float()
    lb = labels.view(-1, 1).to(s.device)
    t = s.gather(1, lb)
    idx = torch.arange(s.shape[1], device=s.device).view(1, -1)
    rank = ((s > t) | ((s == t) & (idx > lb))).sum(dim=1)
    return (rank < k).sum()


def topk_hits(logits, labels, k):
    return int(topk_hits_tensor(logits, labels, k).item())


def evaluate(model, dataset, device, batch_size=64, k=2, num_workers=0):
    device = torch.device(device)
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers,
                                         pin_memory=device.type == "cuda" and num_workers > 0)
    hits = torch.zeros(2, dtype=torch.int64, device=device)
    n = 0
    model.eval()
    with torch.no_grad():
        for img, lb in DevicePrefetcher(loader, device):      # H2D on the copy stream, one batch ahead
            out = model(img)
            hits[0] += topk_hits_tensor(out, lb, 1)
            hits[1] += topk_hits_tensor(out, lb, k)
            n += int(lb.shape[0])
    top1, topk = (int(v) for v in hits.tolist())              # the only device -> host read of the run
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
    ap.add_argument("--num-workers", type=int, default=0, help="DataLoader decode workers")
    ap.add_argument("--uint8-input", action="store_true",
                    help="ship uint8 pixels to the device (4x fewer PCIe bytes) and normalise inside conv1's ingest kernel")
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate on N synthetic test samples")
    args = ap.parse_args(argv)
    labels = args.labels.split(",")
    device = torch.device(args.device)

    model = VGG16(3, len(labels), True, batch_norm=args.batch_norm, fp8_classifier=args.fp8_classifier)
    ckpt = load_snapshot(args.model)
    model.load_state_dict(ckpt["model_state_dict"])
    model = model.to(device)

    synthetic = bool(args.synthetic) or not os.path.isdir(args.data)
    if synthetic:
        n = args.synthetic or 256
        dataset = SyntheticImageDataset(n, len(labels), args.image_size, args.image_size, seed=2, uint8=args.uint8_input)
    else:
        dataset = ExampleDataset(args.data, labels, args.image_size, args.image_size, phase="val", uint8=args.uint8_input)
    if args.uint8_input:
        if synthetic:     # the synthetic generator's affine map (data/synthetic.py: img * 40 + 128)
            model.set_input_normalization(1.0 / 40.0, (128.0 / 40.0,) * 3, (1.0, 1.0, 1.0))
        else:
            model.set_input_normalization(1.0 / 255.0, tuple(IMAGENET_MEAN), tuple(IMAGENET_STD))
    k = min(2, len(labels))
    acc1, acck, n = evaluate(model, dataset, device, args.batch_size, k, num_workers=args.num_workers)
    print("EVALUATION ACCURACY RESULTS: TOP-1=%s%% --- TOP-%d=%s%%" % (acc1 * 100, k, acck * 100))
    return acc1, acck, n


if __name__ == "__main__":
    main()

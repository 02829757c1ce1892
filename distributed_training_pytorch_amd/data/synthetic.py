"""Synthetic image-classification data (there is no network / dataset on the GPU boxes).

* ``SyntheticImageDataset`` — a map-style dataset whose sample ``i`` is a deterministic function
  of ``(seed, i)`` only, so every rank sees the same index->sample map (fixes SURVEY.md §2.6
  quirk #3, the reference's unseeded per-process shuffle at dataset/example_dataset.py:17).
  Samples are *learnable*: class ``c`` adds a fixed low-frequency prototype to the noise, so a
  loss-decreases / accuracy-rises check is meaningful.
* ``DeviceBatchSource`` — an endless iterator of device-resident batches for throughput
  benchmarking (the data loader is out of the timed path, BASELINE.json "synthetic data").
"""
import torch
from torch.utils.data import Dataset

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _prototypes(num_classes, channels, height, width, seed):
    g = torch.Generator().manual_seed(seed * 7919 + 13)
    low = torch.randn(num_classes, channels, 4, 4, generator=g)
    return torch.nn.functional.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)


class SyntheticImageDataset(Dataset):
    def __init__(self, num_samples, num_classes=3, height=224, width=224, channels=3, seed=0,
                 signal=1.0, uint8=False, proto_seed=0):
        self.num_samples = int(num_samples)
        self.num_classes = int(num_classes)
        self.shape = (channels, height, width)
        self.seed = int(seed)
        self.signal = float(signal)
        self.uint8 = bool(uint8)
        # the class prototypes define the TASK and are shared by every split (train / val / test
        # differ in ``seed`` = their noise and sample order only); with per-split prototypes a model
        # trained on one split could never beat chance on another
        self.protos = _prototypes(num_classes, channels, height, width, proto_seed)

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx):
        if idx < 0 or idx >= self.num_samples:
            raise IndexError(idx)
        label = idx % self.num_classes
        g = torch.Generator().manual_seed((self.seed + 1) * 1000003 + idx)
        img = torch.randn(self.shape, generator=g) + self.signal * self.protos[label]
        if self.uint8:
            img = (img * 40 + 128).clamp_(0, 255).to(torch.uint8)
        return img, label


class DeviceBatchSource(object):
    """Pre-generates ``n_distinct`` batches on the device and cycles through them."""

    def __init__(self, batch_size, num_classes=3, height=224, width=224, channels=3, device="cuda",
                 n_distinct=2, seed=0, dtype=torch.float32):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.batches = []
        protos = _prototypes(num_classes, channels, height, width, seed).to(device)
        for _ in range(n_distinct):
            y = torch.randint(0, num_classes, (batch_size,), generator=g)
            s = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g))
            gd = torch.Generator(device=device).manual_seed(s)
            x = torch.randn((batch_size, channels, height, width), generator=gd, device=device)
            y = y.to(device)
            x = (x + protos[y]).to(dtype)
            self.batches.append((x, y))
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return b

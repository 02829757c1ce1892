"""Image-folder classification dataset (``<root>/<label>/<file>``) without cv2 / albumentations.

Parity target: reference ``dataset/example_dataset.py:11-60`` — list files per label directory,
decode to RGB, train pipeline = resize + random rot90 / h-flip / v-flip / blur / median blur /
CLAHE / brightness-contrast / gamma / JPEG compression (each p=0.5) + ImageNet normalise; val
pipeline = resize + normalise; returns ``(float32 CHW tensor, int label)``.

cv2, albumentations and torchvision are not installable here, so decoding uses PIL
and the augmentations are re-implemented on numpy/PIL: rot90, flips, box blur, median filter,
brightness/contrast, gamma and JPEG re-compression are functional equivalents; CLAHE is
approximated by PIL's global ``ImageOps.equalize`` blended 50/50, and the blur / median kernel
ranges (albumentations defaults: blur_limit 3..7) are sampled from {1, 2, 3} box radii and {3, 5}
median sizes ("parity unpinned": albumentations / cv2 cannot be installed here and no reference
fixture pins augmentation outputs, so the transforms are functional stand-ins, not bit matches).  The file order is deterministic and identical on
every rank (quirk #3): shuffling is the sampler's job.
"""
import io
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .synthetic import IMAGENET_MEAN, IMAGENET_STD

IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".ppm", ".webp")


def _pil():
    from PIL import Image, ImageFilter, ImageOps
    return Image, ImageFilter, ImageOps


class ExampleDataset(Dataset):
    def __init__(self, data_path, labels, height, width, phase, seed=0, normalize=True, prob=0.5, uint8=False):
        super().__init__()
        self.data_path = data_path
        self.labels = list(labels)
        self.height, self.width = int(height), int(width)
        self.phase = phase
        self.prob = prob
        self.normalize = normalize
        # uint8=True: return raw CHW uint8 pixels (4x fewer bytes over PCIe); the /255, mean and std
        # are then fused into the first convolution's ingest kernel (ConvFirst.set_input_normalization)
        self.uint8 = bool(uint8)
        self.data_list = self._load_data(data_path, self.labels)
        # augmentation randomness is a function of (seed, epoch, sample index): every rank and every
        # DataLoader worker draws an independent stream for each sample it serves (one shared
        # random.Random(seed) gave all ranks and workers the SAME stream), the run is reproducible
        # whatever the number of workers, and ``set_epoch`` (called by the Trainer next to the
        # sampler's) changes the draws from epoch to epoch
        self.seed = int(seed)
        self.epoch = 0
        self.rng = random.Random(seed)
        self.mean = np.asarray(IMAGENET_MEAN, dtype=np.float32).reshape(3, 1, 1)
        self.std = np.asarray(IMAGENET_STD, dtype=np.float32).reshape(3, 1, 1)

    @staticmethod
    def _load_data(data_path, labels):
        out = []
        for idx, lb in enumerate(labels):
            d = os.path.join(data_path, lb)
            if not os.path.isdir(d):
                raise FileNotFoundError("label directory missing: %s" % d)
            for name in sorted(os.listdir(d)):
                if name.lower().endswith(IMG_EXT):
                    out.append((os.path.join(d, name), idx))
        return out

    def __len__(self):
        return len(self.data_list)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    # ---------------------------------------------------------------- augmentations
    def _augment(self, img):
        Image, ImageFilter, ImageOps = _pil()
        r, p = self.rng, self.prob
        if r.random() < p:
            img = img.rotate(90 * r.randint(0, 3), expand=False) if self.height == self.width else img
        if r.random() < p:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        if r.random() < p:
            img = img.transpose(Image.FLIP_TOP_BOTTOM)
        if r.random() < p:
            img = img.filter(ImageFilter.BoxBlur(r.choice([1, 2, 3])))
        if r.random() < p:
            img = img.filter(ImageFilter.MedianFilter(size=r.choice([3, 5])))
        if r.random() < p:
            img = Image.blend(img, ImageOps.equalize(img), 0.5)
        if r.random() < p:
            a = np.asarray(img, dtype=np.float32)
            alpha = 1.0 + r.uniform(-0.2, 0.2)
            beta = r.uniform(-0.2, 0.2) * 255.0
            a = np.clip((a - a.mean()) * alpha + a.mean() + beta, 0, 255)
            img = Image.fromarray(a.astype(np.uint8))
        if r.random() < p:
            gamma = r.uniform(0.8, 1.2)
            a = np.asarray(img, dtype=np.float32) / 255.0
            img = Image.fromarray((np.power(a, gamma) * 255.0).astype(np.uint8))
        if r.random() < p:
            buf = io.BytesIO()
            img.save(buf, format="JPEG", quality=r.randint(20, 100))
            buf.seek(0)
            img = Image.open(buf).convert("RGB")
        return img

    def __getitem__(self, idx):
        Image, _, _ = _pil()
        path, lb = self.data_list[idx]
        img = Image.open(path).convert("RGB").resize((self.width, self.height), Image.BILINEAR)
        if self.phase == "train":
            self.rng = random.Random((self.seed * 1000003 + self.epoch) * 2654435761 + idx)
            img = self._augment(img)
        if self.uint8:
            return torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1))), lb
        a = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0
        if self.normalize:
            a = (a - self.mean) / self.std
        return torch.from_numpy(np.ascontiguousarray(a)), lb

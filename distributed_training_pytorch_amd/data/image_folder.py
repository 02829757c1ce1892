"""Image-folder classification dataset (``<root>/<label>/<file>``) without cv2 / albumentations.

Parity target: reference ``dataset/example_dataset.py:11-60`` — list files per label directory,
decode to RGB, train pipeline = resize + random rot90 / h-flip / v-flip / blur / median blur /
CLAHE / brightness-contrast / gamma / JPEG compression (each p=0.5) + ImageNet normalise; val
pipeline = resize + normalise; returns ``(float32 CHW tensor, int label)``.

cv2, albumentations and torchvision are not installable here, so decoding uses PIL
and the augmentations are re-implemented on numpy/PIL with the parameter ranges albumentations'
defaults give the reference's calls:

* ``RandomRotate90``: k in {0..3} quarter turns (square targets only: a non-square rotate would
  change the batch shape); ``HorizontalFlip`` / ``VerticalFlip``.
* ``Blur()``: box filter, odd kernel in {3, 5, 7} (blur_limit 3..7) = PIL ``BoxBlur`` radius {1, 2, 3}.
* ``MedianBlur()``: odd kernel in {3, 5, 7}.
* ``CLAHE()``: clip limit uniform in [1, 4], 8x8 tile grid, applied to the luminance — ``clahe()``
  below is the tile-histogram / clip / redistribute / bilinear-LUT algorithm itself (the reference
  gets it from cv2 on the L channel of LAB; here it runs on Y of YCbCr, PIL having no LAB).
* ``RandomBrightnessContrast()``: ``clip(x * (1 + c) + b * 255)``, c, b uniform in [-0.2, 0.2]
  (albumentations' ``brightness_by_max=True`` uint8 path — NOT mean-centred).
* ``RandomGamma()``: gamma uniform in [0.8, 1.2]; ``ImageCompression(20, 100)``: JPEG round trip.

"Parity unpinned": neither library can be installed here and no reference fixture pins an
augmentation output, so these follow the documented algorithms and ranges but are not bit
matches (border handling of the blurs, PIL's antialiased bilinear resize vs cv2 INTER_LINEAR).
The file order is deterministic and identical on every rank (quirk #3): shuffling is the
sampler's job.
"""
import io
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .device_augment import brightness_contrast_u8, gamma_u8
from .synthetic import IMAGENET_MEAN, IMAGENET_STD

IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".ppm", ".webp")


def _pil():
    from PIL import Image, ImageFilter, ImageOps
    return Image, ImageFilter, ImageOps


def clahe(gray, clip_limit=4.0, grid=(8, 8)):
    """Contrast-limited adaptive histogram equalisation of a uint8 [H][W] plane.

    Per tile of the ``grid``: 256-bin histogram, bins clipped at ``clip_limit * tile_pixels / 256``
    (at least 1), the clipped mass redistributed evenly (remainder spread with a stride, as cv2
    does), LUT = round(cdf * 255 / tile_pixels); every pixel is mapped through the four LUTs of
    the tiles around it, blended bilinearly between tile centres.  The plane is edge-padded
    (reflect) to a multiple of the grid first."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    H, W = gray.shape
    gy, gx = int(grid[0]), int(grid[1])
    ph, pw = (-H) % gy, (-W) % gx
    src = np.pad(gray, ((0, ph), (0, pw)), mode="reflect") if (ph or pw) else gray
    th, tw = src.shape[0] // gy, src.shape[1] // gx
    area = th * tw
    limit = max(int(clip_limit * area / 256.0), 1) if clip_limit > 0 else area
    tiles = src.reshape(gy, th, gx, tw).transpose(0, 2, 1, 3).reshape(gy * gx, area)
    hist = np.stack([np.bincount(t, minlength=256) for t in tiles]).astype(np.int64)   # [tiles][256]
    excess = np.maximum(hist - limit, 0).sum(axis=1)
    hist = np.minimum(hist, limit)
    hist += (excess // 256)[:, None]
    residual = excess % 256
    for t in np.nonzero(residual)[0]:
        step = max(256 // int(residual[t]), 1)
        hist[t, np.arange(0, 256, step)[:int(residual[t])]] += 1
    lut = np.clip(np.rint(np.cumsum(hist, axis=1) * (255.0 / area)), 0, 255).astype(np.float32)
    lut = lut.reshape(gy, gx, 256)
    # bilinear blend between tile centres
    yy = (np.arange(H, dtype=np.float32) + 0.5) / th - 0.5
    xx = (np.arange(W, dtype=np.float32) + 0.5) / tw - 0.5
    y0 = np.floor(yy).astype(np.int64); x0 = np.floor(xx).astype(np.int64)
    wy = (yy - y0).astype(np.float32)[:, None]; wx = (xx - x0).astype(np.float32)[None, :]
    y1 = np.clip(y0 + 1, 0, gy - 1)[:, None]; x1 = np.clip(x0 + 1, 0, gx - 1)[None, :]
    y0 = np.clip(y0, 0, gy - 1)[:, None]; x0 = np.clip(x0, 0, gx - 1)[None, :]
    v = gray.astype(np.int64)
    out = ((1 - wy) * ((1 - wx) * lut[y0, x0, v] + wx * lut[y0, x1, v])
           + wy * ((1 - wx) * lut[y1, x0, v] + wx * lut[y1, x1, v]))
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


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
            img = img.filter(ImageFilter.MedianFilter(size=r.choice([3, 5, 7])))
        if r.random() < p:
            ycc = np.array(img.convert("YCbCr"))
            ycc[..., 0] = clahe(ycc[..., 0], clip_limit=r.uniform(1.0, 4.0), grid=(8, 8))
            img = Image.fromarray(ycc, mode="YCbCr").convert("RGB")
        if r.random() < p:
            alpha = 1.0 + r.uniform(-0.2, 0.2)
            beta = r.uniform(-0.2, 0.2) * 255.0
            img = Image.fromarray(brightness_contrast_u8(np.asarray(img), alpha, beta))     # shared with the device pipeline
        if r.random() < p:
            img = Image.fromarray(gamma_u8(np.asarray(img), r.uniform(0.8, 1.2)))
        if r.random() < p:
            buf = io.BytesIO()
            img.save(buf, format="JPEG", quality=r.randint(20, 100))
            buf.seek(0)
            img = Image.open(buf).convert("RGB")
        return img

    def load_raw(self, idx):
        """Decoded, resized, UN-augmented sample as (uint8 CHW tensor, label): what the HBM cache of
        ``data/device_cache.py`` holds; augmentation then runs on the device (``data/device_augment.py``)."""
        Image, _, _ = _pil()
        path, lb = self.data_list[idx]
        img = Image.open(path).convert("RGB").resize((self.width, self.height), Image.BILINEAR)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1))), lb

    def device_augmenter(self):
        """The device-side stand-in for ``_augment`` (rot90 / flips / brightness-contrast / gamma; blur,
        median, CLAHE and JPEG stay CPU-only), drawing from (seed, epoch, index) like the CPU pipeline."""
        from .device_augment import DeviceAugmenter
        return DeviceAugmenter(seed=self.seed, prob=self.prob) if self.phase == "train" else None

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

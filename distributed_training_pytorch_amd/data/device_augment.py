"""Device-side augmentation for HBM-resident image caches (``csrc/augment.hip``).

Parity target: the reference augments every sample on the CPU inside ``__getitem__``
(``dataset/example_dataset.py:35-46``, ``num_workers=0``): a few hundred images/s, while one MI355X
trains VGG16 at ~12,000.  Here the un-augmented uint8 images live in HBM (``data/device_cache.py``)
and ONE kernel per batch gathers the batch and applies, per sample,

  RandomRotate90 -> HorizontalFlip -> VerticalFlip -> RandomBrightnessContrast -> RandomGamma

(each with probability ``prob``, same parameter ranges as ``data/image_folder.py``).  The geometric
part is an index permutation, the photometric part one 256-entry table per sample that the host
composes with the SAME numpy expressions the CPU pipeline uses (:func:`brightness_contrast_u8`,
:func:`gamma_u8` — ``image_folder.ExampleDataset._augment`` calls them too), so for these five
transforms the device result is bit-identical to the CPU pipeline given the same parameters
(``tests/test_gpu_kernels.py::test_device_augmentation_matches_numpy``).

Blur, median blur, CLAHE and the JPEG round trip stay CPU-only: neighbourhood filters / an entropy
codec are not worth a hand-written kernel for a pipeline stand-in ("parity unpinned" either way);
the device pipeline simply does not apply them — ``--device-cache`` trades them for ~40x the input rate.

Randomness is a counter-based function of ``(seed, epoch, sample index, draw number)`` — every
rank, every epoch and every sample gets its own stream and a run is reproducible whatever the
batch composition.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def brightness_contrast_u8(a_u8, alpha, beta):
    """albumentations' RandomBrightnessContrast uint8 path: clip(x * alpha + beta, 0, 255), truncated.
    ``a_u8``: uint8 array; ``alpha`` / ``beta``: scalars (or arrays broadcastable to it)."""
    a = np.asarray(a_u8, dtype=np.float32)
    return np.clip(a * np.float32(alpha) + np.float32(beta), 0, 255).astype(np.uint8)


def gamma_u8(a_u8, gamma):
    """RandomGamma: (x / 255) ** gamma * 255, truncated."""
    a = np.asarray(a_u8, dtype=np.float32) / np.float32(255.0)
    return (np.power(a, np.float32(gamma)) * np.float32(255.0)).astype(np.uint8)


def _hash_uniform(seed, epoch, index, draw):
    """Uniform [0, 1) doubles from a splitmix64 mix of (seed, epoch, index, draw); vectorised over index."""
    with np.errstate(over="ignore"):
        z = (np.asarray(index, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64((int(seed) * 0xD1B54A32D192ED03 + int(epoch) * 0x8CB92BA72F3D8DD7 + int(draw) * 0xABC98388FB8FAC03
                          + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def draw_params(indices, epoch, seed=0, prob=0.5, square=True):
    """Per-sample augmentation parameters for the samples ``indices`` of epoch ``epoch``."""
    idx = np.asarray(indices, dtype=np.int64)
    u = lambda d: _hash_uniform(seed, epoch, idx, d)        # noqa: E731
    k = np.where((u(0) < prob) & square, np.floor(u(1) * 4).astype(np.int32) % 4, 0).astype(np.int32)
    hflip = (u(2) < prob).astype(np.int32)
    vflip = (u(3) < prob).astype(np.int32)
    bc_on = u(4) < prob
    alpha = np.where(bc_on, 1.0 + (u(5) * 0.4 - 0.2), 1.0)
    beta = np.where(bc_on, (u(6) * 0.4 - 0.2) * 255.0, 0.0)
    g_on = u(7) < prob
    gamma = np.where(g_on, 0.8 + u(8) * 0.4, 1.0)
    return dict(k=k, hflip=hflip, vflip=vflip, bc_on=bc_on, alpha=alpha, beta=beta, gamma_on=g_on, gamma=gamma)


def compose_lut(params):
    """[N, 256] uint8: value -> gamma(brightness_contrast(value)) per sample (identity where a
    transform is off), built with the CPU pipeline's own expressions."""
    n = len(params["k"])
    v = np.broadcast_to(np.arange(256, dtype=np.uint8), (n, 256))
    bc = brightness_contrast_u8(v, params["alpha"].astype(np.float32)[:, None], params["beta"].astype(np.float32)[:, None])
    v = np.where(params["bc_on"][:, None], bc, v)
    # gamma on the (possibly changed) values: a second table lookup per sample
    gt = gamma_u8(np.broadcast_to(np.arange(256, dtype=np.uint8), (n, 256)), params["gamma"].astype(np.float32)[:, None])
    g = np.take_along_axis(gt, v.astype(np.int64), axis=1)
    return np.ascontiguousarray(np.where(params["gamma_on"][:, None], g, v).astype(np.uint8))


def geom_codes(params):
    return np.ascontiguousarray((params["k"] | (params["hflip"] << 2) | (params["vflip"] << 3)).astype(np.int32))


def augment_reference(x_u8, params):
    """The same pipeline on the host with numpy ([N, C, H, W] uint8) — what the kernel must equal."""
    out = np.empty_like(x_u8)
    for n in range(x_u8.shape[0]):
        a = x_u8[n]
        a = np.rot90(a, int(params["k"][n]), axes=(1, 2))
        if params["hflip"][n]:
            a = a[:, :, ::-1]
        if params["vflip"][n]:
            a = a[:, ::-1, :]
        if params["bc_on"][n]:
            a = brightness_contrast_u8(a, params["alpha"][n], params["beta"][n])
        if params["gamma_on"][n]:
            a = gamma_u8(a, params["gamma"][n])
        out[n] = a
    return out


class DeviceAugmenter(object):
    def __init__(self, seed=0, prob=0.5):
        self.seed, self.prob = int(seed), float(prob)

    def __call__(self, cache_u8, sel, sample_indices, epoch):
        """``cache_u8`` [n_cached, C, H, W] uint8 on the GPU, ``sel`` int64 device tensor of the rows
        to gather, ``sample_indices`` the dataset indices of those rows (host ints: they key the
        random draws) -> augmented uint8 batch [len(sel), C, H, W]."""
        from .. import _lib
        from .._lib import check, ptr, stream
        N = int(sel.numel())
        _, C, H, W = cache_u8.shape
        p = draw_params(sample_indices, epoch, self.seed, self.prob, square=(H == W))
        dev = cache_u8.device
        geom = torch.from_numpy(geom_codes(p)).to(dev, non_blocking=True)
        lut = torch.from_numpy(compose_lut(p)).to(dev, non_blocking=True)
        out = torch.empty((N, C, H, W), dtype=torch.uint8, device=dev)
        check(_lib.hip().dtp_augment_u8(ptr(cache_u8), ptr(sel), ptr(geom), ptr(lut), ptr(out), N, C, H, W,
                                        1, stream()), "augment_u8")
        return out

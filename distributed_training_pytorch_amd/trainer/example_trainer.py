"""The worked example: VGG16 image classification on an image folder or synthetic data.

Parity target: reference ``example_trainer.py:11-102`` — same constructor
(``train_path, val_path, labels, height, width`` + Trainer kwargs), same nine hooks:
VGG16(3, n_classes), mean cross-entropy, SGD(lr 0.1, momentum 0.9, wd 1e-4),
MultiStepLR([50, 100, 200], 0.1), train_step -> ``{"ce_loss": ...}``, validate_step ->
``{"accuracy": ...}``.

Differences: validation uses ``val_path`` (quirk #2); the optimizer is the fused flat-arena SGD
(one launch); loss / accuracy come from the fused softmax-xent kernel and stay on the device
(no per-step ``.item()``); gradient accumulation via ``grad_accum_steps`` + ``no_sync``;
``train_path="synthetic"`` (or ``synthetic:<n>``) selects the deterministic synthetic dataset
because no real dataset can be downloaded here.
"""
import contextlib

import torch

from ..data.image_folder import ExampleDataset
from ..data.synthetic import SyntheticImageDataset
from ..models.vgg16 import VGG16
from ..ops.layers import SoftmaxCrossEntropy, top1_hits
from ..ops.optim import FusedSGD
from ..utils.profiler import trace_range
from .trainer import Trainer


def _synthetic_len(path, default):
    if isinstance(path, str) and path.startswith("synthetic"):
        parts = path.split(":")
        return int(parts[1]) if len(parts) > 1 and parts[1] else default
    return None


class ExampleTrainer(Trainer):
    def __init__(self,
                 train_path,
                 val_path,
                 labels,
                 height,
                 width,
                 max_epoch,
                 batch_size,
                 pin_memory,
                 have_validate=False,
                 save_best_for=None,
                 save_period=None,
                 save_folder='.',
                 snapshot_path=None,
                 logger=None,
                 lr=0.1,
                 momentum=0.9,
                 weight_decay=1e-4,
                 milestones=(50, 100, 200),
                 batch_norm=False,
                 grad_dtype=torch.float32,
                 fp8_classifier=False,
                 uint8_input=False,
                 **engine_kwargs):
        self.train_path = train_path
        self.val_path = val_path
        self.labels = labels
        self.height = height
        self.width = width
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.milestones = list(milestones)
        self.batch_norm = batch_norm
        self.grad_dtype = grad_dtype
        self.fp8_classifier = fp8_classifier
        # ship uint8 pixels host->device and normalise inside the conv1 ingest kernel
        self.uint8_input = bool(uint8_input)
        super().__init__(max_epoch, batch_size, pin_memory, have_validate, save_best_for, save_period,
                         save_folder, snapshot_path, logger, **engine_kwargs)

    # datasets -------------------------------------------------------------------------------
    def _dataset(self, path, phase, default_len, seed):
        n = _synthetic_len(path, default_len)
        if n is not None:
            return SyntheticImageDataset(n, len(self.labels), self.height, self.width, seed=seed,
                                         uint8=self.uint8_input)
        return ExampleDataset(path, self.labels, self.height, self.width, phase=phase, uint8=self.uint8_input)

    def build_train_dataset(self):
        return self._dataset(self.train_path, "train", 1024, seed=0)

    def build_val_dataset(self):
        return self._dataset(self.val_path, "val", 256, seed=1)

    # model / loss / optimiser -----------------------------------------------------------------
    def build_model(self):
        model = VGG16(3, len(self.labels), True, batch_norm=self.batch_norm, grad_dtype=self.grad_dtype,
                      fp8_classifier=self.fp8_classifier)
        if self.uint8_input:
            model.set_input_normalization(*self.input_normalization())
        return model

    def input_normalization(self):
        """(scale, mean, std) that turn the dataset's uint8 pixels into what the float pipeline
        feeds the network: ImageNet statistics for image folders, the generator's affine map for
        the synthetic set."""
        if _synthetic_len(self.train_path, 1) is not None:
            return 1.0 / 40.0, (128.0 / 40.0,) * 3, (1.0, 1.0, 1.0)
        from ..data.image_folder import IMAGENET_MEAN, IMAGENET_STD
        return 1.0 / 255.0, tuple(IMAGENET_MEAN), tuple(IMAGENET_STD)

    def build_criterion(self):
        return SoftmaxCrossEntropy()

    def build_optimizer(self):
        return FusedSGD(self.model, lr=self.lr, momentum=self.momentum, weight_decay=self.weight_decay)

    def build_scheduler(self):
        return torch.optim.lr_scheduler.MultiStepLR(self.optimizer, self.milestones, gamma=0.1)

    # steps ----------------------------------------------------------------------------------
    def preprocess_batch(self, batch):
        return [e.to(self.device, non_blocking=True) if torch.is_tensor(e) else e for e in batch]

    def train_step(self, batch):
        img_batch, lb_batch = self.preprocess_batch(batch)
        if self.graphed_step is not None:
            # zero_grad + forward + loss + backward + step as ONE hipGraph launch
            return dict(ce_loss=self.graphed_step(img_batch, lb_batch))
        micro = getattr(self, "micro_step", 0)
        first = micro % self.grad_accum_steps == 0
        sync = self.should_sync(micro)           # also True on the epoch's last step: short windows are flushed
        window = self.accum_window(micro)        # micro-steps in this window (== grad_accum_steps except at the tail)
        if first:
            self.optimizer.zero_grad()
        ctx = contextlib.nullcontext() if sync else self.model.no_sync()
        timer = self.step_timer
        with ctx, torch.set_grad_enabled(True):
            with trace_range("fwd"), timer.phase("fwd"):
                out = self.model(img_batch)
                loss = self.criterion(out, lb_batch)
            with trace_range("bwd"), timer.phase("bwd"):
                (loss / window if window > 1 else loss).backward()
        if sync:
            with trace_range("optimizer"), timer.phase("optimizer"):
                self.optimizer.step()
        return dict(ce_loss=loss.detach())

    def validate_step(self, batch):
        img_batch, lb_batch = self.preprocess_batch(batch)
        with torch.set_grad_enabled(False):
            logits = self.raw_model(img_batch)       # raw module: no collectives in eval (quirk #9)
            acc = top1_hits(logits, lb_batch) / lb_batch.shape[0]
        return dict(accuracy=acc)

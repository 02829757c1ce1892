"""Compatibility import path of the reference (``from model.vgg16 import VGG16``,
reference example_trainer.py:8, eval.py:12)."""
from distributed_training_pytorch_amd.models.vgg16 import VGG16, ConvBlock, vgg16_bn  # noqa: F401

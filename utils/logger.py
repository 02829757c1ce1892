"""Compatibility import path of the reference (``from utils.logger import Logger``,
reference main.py:2)."""
from distributed_training_pytorch_amd.utils.logger import Logger  # noqa: F401

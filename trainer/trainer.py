"""Compatibility import path of the reference (``from trainer.trainer import Trainer``,
reference example_trainer.py:6).  The implementation lives in the package."""
from distributed_training_pytorch_amd.trainer.trainer import Trainer  # noqa: F401

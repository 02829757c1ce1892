"""Compatibility import path of the reference (``from example_trainer import ExampleTrainer``,
reference main.py:1)."""
from distributed_training_pytorch_amd.trainer.example_trainer import ExampleTrainer  # noqa: F401

"""Compatibility import path of the reference (``from dataset.example_dataset import
ExampleDataset``, reference example_trainer.py:7)."""
from distributed_training_pytorch_amd.data.image_folder import ExampleDataset  # noqa: F401
from distributed_training_pytorch_amd.data.synthetic import SyntheticImageDataset  # noqa: F401

from .graph import GraphedTrainStep  # noqa: F401

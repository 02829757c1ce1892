from .fused_model import FusedModel, attach_arena  # noqa: F401
from .param_arena import ParamArena  # noqa: F401
from .vgg16 import VGG16, ConvBlock, vgg16_bn  # noqa: F401

from .encoder import Encoder, EncLayer  # noqa: F401
from .mlp import MLP  # noqa: F401
from . import common, sparse, uniform, extension  # noqa: F401

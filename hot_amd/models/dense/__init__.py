from .linear import Linear  # noqa: F401
from .kernelattn import KernelSelfAttn  # noqa: F401
from .attn import SelfAttn  # noqa: F401
from .f import Apply, Nonlinear, Add  # noqa: F401

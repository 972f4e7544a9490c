from .kernel import KernelFeatureMap  # noqa: F401
from .mudrop import MuDropout  # noqa: F401

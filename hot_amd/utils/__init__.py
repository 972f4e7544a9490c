from . import set, sparse  # noqa: F401

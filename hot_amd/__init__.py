"""hot_amd: MI355X-native (gfx950) drop-in for the second-order sparse kernel-attention path of
hot_pytorch (jw9730/hot).  Same Batch object, same layer constructors and forward signatures; the work is
done by hand-written HIP kernels in libhotk.so reached through a C ABI (include/hotk.h)."""
from . import batch, models, utils  # noqa: F401

__all__ = ['batch', 'models', 'utils']

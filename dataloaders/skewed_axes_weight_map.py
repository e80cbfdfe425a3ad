from distributedpytorch_amd.data.skewed_axes_weight_map import *  # noqa: F401,F403
from distributedpytorch_amd.data import skewed_axes_weight_map as _impl  # noqa: F401
__all__ = getattr(_impl, '__all__', [n for n in dir(_impl) if not n.startswith('_')])

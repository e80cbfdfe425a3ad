from distributedpytorch_amd.data.helpers import *  # noqa: F401,F403
from distributedpytorch_amd.data import helpers as _impl  # noqa: F401
__all__ = getattr(_impl, '__all__', [n for n in dir(_impl) if not n.startswith('_')])

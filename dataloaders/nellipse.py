from distributedpytorch_amd.data.nellipse import *  # noqa: F401,F403
from distributedpytorch_amd.data import nellipse as _impl  # noqa: F401
__all__ = getattr(_impl, '__all__', [n for n in dir(_impl) if not n.startswith('_')])

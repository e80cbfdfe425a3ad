from distributedpytorch_amd.data.transforms import *  # noqa: F401,F403

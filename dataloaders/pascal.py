from distributedpytorch_amd.data.pascal import *  # noqa: F401,F403
from distributedpytorch_amd.data.pascal import VOCSegmentation, SyntheticVOC  # noqa: F401

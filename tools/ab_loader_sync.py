#!/usr/bin/env python
"""A/B on one box: bench.py with a full device synchronisation put back into the instance loader (what the blocking
identity-matrix copy did until round 3) against the loader as it is.  `python tools/ab_loader_sync.py sync|nosync <bench args>`; modes aux-normal / aux-low: the same for the priority of
the weight-gradient stream."""
import os
import runpy
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "sync":
    import distributedpytorch_amd.ops.augment as aug
    orig = aug._identity_mats

    def synced(n, device):
        torch.cuda.synchronize()
        return orig(n, device)
    aug._identity_mats = synced
if mode == "aux-normal":            # weight-gradient stream at normal priority (a torch pool stream)
    from distributedpytorch_amd.ops import streams
    streams._PRIORITY = "normal"
if mode == "aux-low":
    from distributedpytorch_amd.ops import streams
    streams._PRIORITY = "low"
if mode.startswith("reps"):           # cap of the statistics replicas (default 8): repsN
    from distributedpytorch_amd.ops import _hip_conv
    cap = int(mode[4:])
    _hip_conv.stat_replicas = lambda m_rows: max(1, min(cap, ((m_rows + 127) // 128) // 64))
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")

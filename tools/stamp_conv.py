#!/usr/bin/env python
"""Where does a conv_igemm2 block spend its life?  Diagnostic build of the kernel library with -DDP_STAMP
(csrc/conv_igemm2.hip: seven s_memtime stamps per block, never in libdpamd_hip.so) and a per-phase average.

    python tools/stamp_conv.py --build                        # on the CPU box: csrc/build/libdpamd_stamp.so
    python tools/stamp_conv.py --batch 64 --cfgs 5,9 728:728:65:1 256:1024:33:1     # on the GPU

phases:  setup (entry -> first DMA issued) | fill (first stage landed) | kloop | stage C (acc -> LDS) |
         store loop issued | stores drained
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "distributedpytorch_amd", "csrc")
STAMP_LIB = os.path.join(CSRC, "build", "libdpamd_stamp.so")


def build(lib_path, defs):
    from distributedpytorch_amd.csrc import build as b
    b.build_all()
    objs = []
    for src in b.hip_sources():
        if os.path.basename(src) in ("conv_igemm2.hip", "conv_wgrad.hip"):
            obj = lib_path[:-3] + "." + os.path.basename(src) + ".o"
            subprocess.check_call([b.HIPCC, *b.HIP_FLAGS, "-DDP_STAMP=1", *[f"-D{d}" for d in defs], "-c", src, "-o", obj])
        else:
            obj = os.path.join(b.OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
    objs.append(os.path.join(b.OBJ_DIR, "_build_stamp.o"))
    subprocess.check_call([b.HIPCC, "-shared", "-fPIC", f"--offload-arch={b.ARCH}", "-o", lib_path, *objs])
    print("built", lib_path)


def wgrad_main(a, lib, dbg, _hip_conv, torch):
    """phases of a weight-gradient block: setup | fill | K loop (pixel steps) | - | atomics issued | atomics drained"""
    lib.dp_wgrad_set_dbg.argtypes = [ctypes.c_void_p]
    names = ["setup", "fill", "kloop", "-", "atomics", "drain"]
    print(f"{'shape':20s} {'blocks':>7s} {'steps':>6s} {'us':>8s} {'TF':>6s} | " + " ".join(f"{n:>7s}" for n in names)
          + " |   total  (us per block, mean over blocks; clock GHz)")
    for sh in a.shapes:
        f = [int(v) for v in sh.split(":")]
        cin, cout, hw, k = f[:4]
        dil = f[4] if len(f) > 4 else 1
        pad = dil * (k - 1) // 2
        n = a.batch
        cl = torch.channels_last
        x = torch.randn(n, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
        dy = torch.randn(n, cout, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
        dw = torch.zeros(cout, cin, k, k, device="cuda").contiguous(memory_format=cl)
        gf = 2.0 * n * hw * hw * cin * cout * k * k / 1e9

        def run():
            _hip_conv.wgrad(dy, x, (cout, cin, k, k), (1, 1), (pad, pad), (dil, dil), out=dw)
        lib.dp_wgrad_set_dbg(None)
        for _ in range(3):
            run()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        t_us = sorted(ts)[2]
        dbg.zero_()
        lib.dp_wgrad_set_dbg(ctypes.c_void_p(dbg.data_ptr()))
        run()
        torch.cuda.synchronize()
        lib.dp_wgrad_set_dbg(None)
        d = dbg.view(-1, 16).cpu()
        d = d[d[:, 0] != 0].double()
        if d.numel() == 0:
            print(f"{sh:20s} (no stamps: first-generation kernel)")
            continue
        ghz = float(((d[:, 6] - d[:, 0]) / (d[:, 8] - d[:, 7]).clamp(min=1)).median()) * 0.1
        ph = (d[:, 1:7] - d[:, 0:6]).mean(0) / (ghz * 1e3)
        tot = float((d[:, 6] - d[:, 0]).mean() / (ghz * 1e3))
        print(f"{sh:20s} {d.shape[0]:7d} {float(d[:, 11].mean()):6.1f} {t_us:8.1f} {gf / t_us * 1e3:6.0f} | "
              + " ".join(f"{float(v):7.2f}" for v in ph) + f" | {tot:7.2f}  ({ghz:.2f})", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--lib", default=STAMP_LIB, help="diagnostic library to build / load")
    ap.add_argument("--defs", default="", help="';'-separated extra -D definitions of the diagnostic build")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--cfgs", default="5")
    ap.add_argument("--modes", default="plain,stats,res,resaff,bn")
    ap.add_argument("--wgrad", action="store_true", help="stamp the weight-gradient kernel instead (csrc/conv_wgrad.hip)")
    ap.add_argument("shapes", nargs="*", help="cin:cout:hw:k[:dil]")
    a = ap.parse_args()
    if a.build:
        return build(os.path.abspath(a.lib), [d for d in a.defs.split(";") if d])
    import torch
    from distributedpytorch_amd.ops import _backend as B
    B._lib = ctypes.CDLL(os.path.abspath(a.lib))            # the diagnostic library stands in for libdpamd_hip.so
    from distributedpytorch_amd.ops import _hip_conv
    lib = _hip_conv._lib()
    lib.dp_conv_set_dbg.argtypes = [ctypes.c_void_p]
    staggers = [(0, 0)]            # (the start-time stagger experiment is concluded and its code removed: README)
    dbg = torch.zeros(16 * (1 << 19), dtype=torch.int64, device="cuda")
    if a.wgrad:
        return wgrad_main(a, lib, dbg, _hip_conv, torch)
    names = ["setup", "fill", "kloop", "stageC", "store", "drain"]
    print(f"{'shape':20s} {'mode':7s} cfg {'blocks':>7s} {'us':>8s} {'TF':>6s} | " + " ".join(f"{n:>7s}" for n in names)
          + " |   total  (us per block, mean over blocks; clock GHz; blocks of a CU inside their K loops at once)")
    for sh in a.shapes:
        f = [int(v) for v in sh.split(":")]
        cin, cout, hw, k = f[:4]
        dil = f[4] if len(f) > 4 else 1
        pad = dil * (k - 1) // 2
        n = a.batch
        x = torch.randn(n, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).to(torch.bfloat16) \
            .contiguous(memory_format=torch.channels_last)
        res = torch.randn(n, cout, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ra = torch.randn(2, cout, device="cuda")
        reps = _hip_conv.stat_replicas(n * hw * hw)
        stats = torch.zeros(reps, 2, cout, device="cuda")
        holder = {"x": res, "mean": torch.zeros(cout, device="cuda"), "invstd": torch.ones(cout, device="cuda"),
                  "w": torch.ones(cout, device="cuda"), "b": torch.zeros(cout, device="cuda")}
        gf = 2.0 * n * hw * hw * cin * cout * k * k / 1e9
        for cfg in [int(c) for c in a.cfgs.split(",")]:
          for mode in a.modes.split(","):
            for stg in staggers:
                def run():
                    if mode == "plain":
                        _hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=cfg)
                    elif mode == "stats":
                        _hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=cfg, stats=stats)
                    elif mode == "res":
                        _hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=cfg, residual=res)
                    elif mode == "resaff":
                        _hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=cfg, residual=res,
                                        residual_affine=ra)
                    elif mode == "bn":
                        prev = _hip_conv._FORCED_CFG
                        _hip_conv._FORCED_CFG = cfg
                        try:
                            _hip_conv.dgrad_bn_fused(x, w, (hw, hw), 1, (pad, pad), (dil, dil), res, holder)
                        finally:
                            _hip_conv._FORCED_CFG = prev
                lib.dp_conv_set_dbg(None)
                for _ in range(3):
                    run()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(); e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                t_us = sorted(ts)[2]
                dbg.zero_()
                lib.dp_conv_set_dbg(ctypes.c_void_p(dbg.data_ptr()))
                run()
                torch.cuda.synchronize()
                lib.dp_conv_set_dbg(None)
                d = dbg.view(-1, 16).cpu()
                d = d[d[:, 0] != 0].double()
                if d.numel() == 0:
                    print(f"{sh:20s} {mode:7s} {cfg:3d}  (no stamps: shape not taken)")
                    continue
                # shader clock: s_memtime ticks per 100 MHz s_memrealtime tick, median over blocks
                ghz = float(((d[:, 6] - d[:, 0]) / (d[:, 8] - d[:, 7]).clamp(min=1)).median()) * 0.1
                ph = (d[:, 1:7] - d[:, 0:6]).mean(0) / (ghz * 1e3)
                tot = float((d[:, 6] - d[:, 0]).mean() / (ghz * 1e3))
                # how many blocks of one CU sit in their K loops together: sum of K-loop spans / length of their union
                hwid = d[:, 9].long()
                key = (d[:, 10].long() & 15) * 4096 + ((hwid >> 8) & 0xff)      # XCC, SE|SH|CU
                conc = []
                for kv in key.unique()[:64]:
                    iv = d[key == kv][:, 2:4]
                    iv = iv[iv[:, 0].argsort()]
                    union, end = 0.0, -1.0
                    for b0, b1 in iv.tolist():
                        if b0 > end:
                            union += b1 - b0
                        elif b1 > end:
                            union += b1 - end
                        end = max(end, b1)
                    conc.append(float((iv[:, 1] - iv[:, 0]).sum()) / max(union, 1.0))
                cc = sum(conc) / max(len(conc), 1)
                tag = f"{mode}" + ("" if stg == (0, 0) else f"+{stg[0]},{stg[1]}")
                print(f"{sh:20s} {tag:12s} {cfg:3d} {d.shape[0]:7d} {t_us:8.1f} {gf / t_us * 1e3:6.0f} | "
                      + " ".join(f"{float(v):7.2f}" for v in ph) + f" | {tot:7.2f}  ({ghz:.2f}; {cc:.2f} of {len(key.unique())} CUs)",
                      flush=True)


if __name__ == "__main__":
    main()

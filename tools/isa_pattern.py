#!/usr/bin/env python
"""Compressed instruction pattern of a kernel's ISA: where does it wait?  Compiles one csrc/*.hip with -save-temps and
prints, per kernel matching the regex, the sequence of MFMAs (M), LDS reads / writes (r / w), global loads / stores /
atomics (g / s / a), barriers and s_waitcnt (W<lgkmcnt> / Wv<vmcnt>) with run lengths — `r2 W0 M r2 W0 M` is an LDS
round trip per MFMA, `g Wv0 g Wv0` a chain of dependent global loads.

    python tools/isa_pattern.py attention.hip pam_bwd
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from distributedpytorch_amd.csrc import build as b
    src = os.path.join(b.HERE, sys.argv[1])
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else re.compile(".")
    with tempfile.TemporaryDirectory() as td:
        subprocess.run([b.HIPCC, *b.HIP_FLAGS, "-save-temps=obj", "-c", src, "-o", os.path.join(td, "x.o")],
                       check=True, capture_output=True)
        asm = open([f for f in glob.glob(os.path.join(td, "*gfx950*.s"))][0]).read()
    for m in re.finditer(r"^(_Z\S+):", asm, re.M):
        name = m.group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if not pat.search(dem) or ".end_amdhsa_kernel" not in asm[m.end():]:
            continue
        body = asm[m.end():asm.index(".end_amdhsa_kernel", m.end())]
        if "s_endpgm" not in body:
            continue
        ev = []
        for l in body.split("\n"):
            l = l.strip()
            if l.startswith("v_mfma"): ev.append("M")
            elif l.startswith("ds_read") or l.startswith("ds_load"): ev.append("r")
            elif l.startswith("ds_write") or l.startswith("ds_store"): ev.append("w")
            elif l.startswith("s_waitcnt"):
                a = re.search(r"lgkmcnt\((\d+)\)", l); v = re.search(r"vmcnt\((\d+)\)", l)
                ev.append("W" + (a.group(1) if a else "") + ("v" + v.group(1) if v else ""))
            elif l.startswith("s_barrier"): ev.append("|B|")
            elif re.match(r"(global|buffer|flat)_load", l): ev.append("g")
            elif re.match(r"(global|buffer|flat)_store", l): ev.append("s")
            elif re.match(r"(global|buffer|flat)_atomic", l): ev.append("a")
            elif l.startswith("scratch_"): ev.append("SCR")
            elif l.startswith("s_cbranch"): ev.append("J")
        out, prev, c = [], None, 0
        for e in ev + [None]:
            if e == prev:
                c += 1
            else:
                if prev:
                    out.append(prev + (str(c) if c > 1 else ""))
                prev, c = e, 1
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", asm[m.end():])
        print(f"== {dem[:110]}  (vgpr {vg.group(1) if vg else '?'})\n" + " ".join(out) + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Summarise a rocprofv3 ``*_kernel_stats.csv`` into a short markdown table grouped by kernel family.

usage: python tools/summarize_prof.py <kernel_stats.csv> <steps> [title] > profiles/<name>.md
"""
import csv
import re
import sys

FAMILIES = [
    # (own kernels first: `dwconv_bwd_data_kernel` would otherwise match the library pattern `conv_bwd_data`)
    ("dpamd depthwise conv", r"dwconv"),
    ("conv fwd (MIOpen igemm)", r"igemm_fwd"),
    ("conv dgrad (MIOpen/CK)", r"igemm_bwd|conv_bwd_data"),
    ("conv wgrad (MIOpen/CK)", r"igemm_wrw|conv_bwd_weight"),
    ("dpamd conv wgrad", r"conv_wgrad"),
    ("dpamd conv fwd/dgrad igemm", r"dp_conv|igemm_kernel|conv_igemm"),
    ("dpamd batchnorm", r"bn_stats_kernel|bn_apply_kernel|bn_bwd|bn_fold"),
    ("dpamd attention", r"pam_|cam_"),
    ("dpamd data path", r"affine_augment|mask_bbox|compose_crop|extreme_points|nellipse"),
    ("dpamd misc", r"dp_|softmax_ce|upsample_ce|bilinear_|maxpool|avgpool|sgd_flat|affine_aug|confusion|dwconv|weight_transpose|pad_channels|p2p_"),
    ("MIOpen batchnorm", r"MIOpenBatchNorm"),
    ("upsample (ATen)", r"upsample_bilinear"),
    ("softmax/nll (ATen)", r"SoftMax|nll_loss"),
    ("elementwise/copy (ATen)", r"elementwise|copy|fillBuffer|SubTensorOp|CatArray|reduce_kernel"),
    ("GEMM (hipBLASLt/rocBLAS)", r"Cijk_|gemm"),
]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else path
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = {}
    for r in rows:
        name = r["Name"]
        key = "other"
        for label, pat in FAMILIES:
            if re.search(pat, name):
                key = label
                break
        t, c = fam.get(key, (0.0, 0))
        fam[key] = (t + float(r["TotalDurationNs"]), c + int(r["Calls"]))
    print(f"# {title}\n")
    print(f"rocprofv3 --kernel-trace --stats; {steps} profiled steps; "
          f"total kernel time {total / 1e6:.1f} ms = {total / 1e6 / steps:.2f} ms/step\n")
    print("| family | ms/step | % | launches/step |\n|---|---:|---:|---:|")
    for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"| {k} | {t / 1e6 / steps:.2f} | {100 * t / total:.1f} | {c / steps:.0f} |")
    print("\n## top 25 kernels\n\n| ms/step | % | calls/step | avg µs | kernel |\n|---:|---:|---:|---:|---|")
    for r in rows[:25]:
        t = float(r["TotalDurationNs"])
        print(f"| {t / 1e6 / steps:.2f} | {100 * t / total:.1f} | {int(r['Calls']) / steps:.0f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | `{r['Name'][:100]}` |")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Merge several autotune result files (DPAMD_CONV_TUNE_FILE of separate runs) into the packaged table
``distributedpytorch_amd/ops/tuned_gfx950.json``: per problem the majority choice, the first file's on a tie.

    python tools/merge_tune.py out.json tune1.json tune2.json tune3.json
"""
import collections
import json
import sys


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    tables = [json.load(open(p)) for p in paths]
    merged, split = {}, 0
    for key in tables[0]:
        votes = [t[key] for t in tables if key in t]
        best = collections.Counter(votes).most_common(1)[0]
        merged[key] = best[0] if best[1] > 1 or len(votes) == 1 else votes[0]
        split += len(set(votes)) > 1
    with open(out, "w") as f:
        json.dump(merged, f, indent=0, sort_keys=True)
    print(f"{len(merged)} problems, {split} with differing choices between runs", file=sys.stderr)
    print(collections.Counter(v for k, v in merged.items() if not k.startswith('["wgrad"')), file=sys.stderr)


if __name__ == "__main__":
    main()

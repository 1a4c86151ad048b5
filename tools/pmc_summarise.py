#!/usr/bin/env python3
"""Aggregate tools/pmc_collect.sh output into per-kernel, per-launch means (JSON on stdout)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def main(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*", "p_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::|x2i_gemm::", "", r["Kernel_Name"])
            if not ("gemm" in name or "attn" in name or "conv5x5" in name):
                continue
            key = "%s grid=%s" % (name.split("(")[0].replace("void ", ""), r["Grid_Size"])
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in agg.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_corrected"] = 2.0 * e["FETCH_SIZE"] * 1024  # gfx950: FETCH_SIZE counts 64 B per 128 B request
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        out[k] = e
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])

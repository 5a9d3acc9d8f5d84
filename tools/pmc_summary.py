#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import collections
import csv
import sys


def main(path, only="k_"):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    for r in rows:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("rcfm::", "").replace("fftk::", "")
        k = k.split("(")[0].replace("void ", "")[:100]
        if only not in k:
            continue
        key = (k, r.get("Grid_Size"))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
    for key, v in agg.items():
        print(key)
        for c, val in sorted(v.items()):
            print("   %-28s %16.0f" % (c, val / cnt[key][c]))


if __name__ == "__main__":
    main(*sys.argv[1:])

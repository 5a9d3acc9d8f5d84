#!/usr/bin/env python3
"""Join a per-step kernel table (rocpd_summary.py --per-step) with a per-kernel traffic table (traffic_summary.py):
kernel, us per launch, bytes read / written at the memory side, TB/s -- the path's kernels against the ~5 TB/s they
converge to.   tools/kernel_rates.py profiles/<tag>_kernel_stats.md profiles/<tag>_hbm_traffic.md (CPU)"""
import re
import sys


def key(name):
    name = name.replace("rcfm::", "").replace("fftk::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"(k_[a-z0-9_]+)<([^>]*(?:<[^>]*>[^>]*)*)", name)
    if not m:
        return name[:40]
    args = re.sub(r"\s+", "", m.group(2))
    return (m.group(1) + "<" + args)[:70]


def main(stats, traffic):
    times = {}
    for line in open(stats):
        m = re.match(r"\| `(.*?)`? \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m and "launches per step" not in line:
            times.setdefault(key(m.group(1)), (m.group(1), float(m.group(3))))   # us per launch (traffic is per launch too)
        if line.startswith("Every kernel"):
            break
    rows = []
    for line in open(traffic):
        m = re.match(r"\| `(.*?)`? \(\d+\) \| ([\d.e+]+) \| .*? \| \d+ \| [\d.]+ \| ([\d.e+]+) \|", line)
        if m:
            k = key(m.group(1))
            if k in times:
                rows.append((times[k][0], times[k][1], float(m.group(2)), float(m.group(3))))
    print("| kernel | us per launch | read GB | written GB | TB/s |")
    print("|---|---:|---:|---:|---:|")
    tot_t = tot_b = 0.0
    for name, us, rd, wr in sorted(rows, key=lambda r: -r[1]):
        print("| `%s` | %.1f | %.3f | %.3f | %.2f |" % (name[:90], us, rd / 1e9, wr / 1e9, (rd + wr) / us / 1e6))
        tot_t += us
        tot_b += rd + wr
    print("| **one launch of each (= one buffer)** | %.1f | | %.2f GB moved | %.2f |" % (tot_t, tot_b / 1e9, tot_b / tot_t / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

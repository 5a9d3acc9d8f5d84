#!/usr/bin/env python3
"""Dump the kernel statistics of a rocprofv3 rocpd database (or --stats CSV dir) as a
markdown table: name, calls, total ms, average us, % -- the summary kept under profiles/."""

import sqlite3
import sys


def per_step(path, steps):
    """Only the library's kernels (rcfm::), as time per hot-path step: the traced process ran `steps` steps of the
    path (warm-up + profile pass + timed steps), so calls / steps = launches per step and total / steps = us per step.
    Input synthesis (torch, rocFFT) and the other configurations are left out."""
    db = sqlite3.connect(path)
    rows = [r for r in db.execute("select name, total_calls, total_duration, average from top_kernels") if "rcfm::" in r[0]]
    total = sum(r[2] for r in rows)
    print("rcfm:: kernels only; %d steps of the path in the traced process; sum = %.1f us per step" % (steps, total / steps))
    print()
    print("| kernel | launches per step | us per launch | us per step | % of the step |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, tot, avg in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rcfm::fftk::", "").replace("rcfm::", "")
        if len(short) > 100:
            short = short[:97] + "..."
        print("| `%s` | %.2f | %.2f | %.1f | %.1f |" % (short, calls / steps, avg, tot / steps, 100.0 * tot / total))


def main(path, limit=40, only=None):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in rows[:limit]:
        if only and only not in name:
            continue
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if len(short) > 110:
            short = short[:107] + "..."
        print("| `%s` | %d | %.3f | %.2f | %.2f |" % (short, calls, total / 1e3, avg, pct))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--per-step":
        per_step(sys.argv[1], int(sys.argv[3]))
        sys.exit(0)
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, sys.argv[3] if len(sys.argv) > 3 else None)

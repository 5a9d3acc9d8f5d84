#!/usr/bin/env python3
"""Dump the kernel statistics of a rocprofv3 rocpd database (or --stats CSV dir) as a
markdown table: name, calls, total ms, average us, % -- the summary kept under profiles/."""

import sqlite3
import sys


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
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, sys.argv[3] if len(sys.argv) > 3 else None)

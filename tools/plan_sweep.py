#!/usr/bin/env python3
"""Times every ordered three-pass plan of the FFT engine for one length (GPU box): the check that the planner's cost
model picks a plan within a few percent of the best one.   python tools/plan_sweep.py 10000000
(N = 10^7: 200.250.200 chosen, 0.105 ms, best 200.625.80 at 0.103; N = 10^8: 400.625.400 chosen and best, 1.075 ms.)"""
import itertools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENGTHS = [75, 80, 100, 120, 125, 128, 150, 160, 192, 200, 240, 250, 256, 300, 320, 375, 384, 400, 480, 500, 512, 600, 625, 640]


def main(n):
    plans = [p for p in itertools.product(LENGTHS, repeat=3) if p[0] * p[1] * p[2] == n]
    res = []
    for p in plans:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_fft.py"), "%d:1" % n,
                              "--plan", ",".join(map(str, p))], capture_output=True, text=True).stdout
        ms = [float(line.split()[3]) for line in out.splitlines() if "engine" in line]
        res.append((ms[0] if ms else float("inf"), p))
        print(p, ms, flush=True)
    print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_fft.py"), "%d:1" % n],
                         capture_output=True, text=True).stdout)
    print("best:", sorted(res)[:6])


if __name__ == "__main__":
    main(int(sys.argv[1]))

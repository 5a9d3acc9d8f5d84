#!/usr/bin/env python3
"""Per-launch table of ONE demodulator call from a rocprofv3 kernel trace: for every kernel of the call its duration on
the device and the idle gap since the previous kernel ended -- what a launch-bound chain is made of (GPU box).

    tools/launch_gaps.py <kind: WBFM|MFM|FM> [synchronous: 0|1]      (re-executes itself under rocprofv3)

The traced process runs 40 calls (240 000 -> 48 000, device in / out, plain launches); the table is the median over the
last 30 calls, kernel by kernel in launch order.  `device busy` = sum of durations, `chain` = first start .. last end."""
import csv
import glob
import os
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(kind, sync):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
    import ctypes
    import torch
    import workloads
    from radiocore._internal import hip
    lib = hip.lib()
    B, A = 240000, 48000
    k = {"FM": 0, "MFM": 1, "WBFM": 2}[kind]
    x = hip.to_device(workloads.single_channel(B, i=0, stereo=(kind == "WBFM")), torch.complex64)
    y = torch.empty(A, 2 if k == 2 else 1, device="cuda")
    h = ctypes.c_void_p()
    hip.check(lib.rcfm_demod_create(k, 1, B, A, ctypes.c_double(75e-6), 0, ctypes.byref(h)))
    s = hip.stream()
    for _ in range(40):
        hip.check(lib.rcfm_demod_run(h, 0, 1, hip.ptr(x), hip.ptr(y), s))
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    hip.check(lib.rcfm_demod_destroy(h))


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "--child":
        return child(sys.argv[1], int(sys.argv[2]))
    kind = sys.argv[1] if len(sys.argv) > 1 else "WBFM"
    sync = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    d = tempfile.mkdtemp(prefix="gaps_")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "-d", d, "-o", "run", "--output-format", "csv", "--", sys.executable,
                    os.path.abspath(__file__), kind, str(sync), "--child"], cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel trace produced")
        return
    rows = [r for r in csv.DictReader(open(files[0])) if "rcfm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    per_call = len(rows) // 40
    assert per_call * 40 == len(rows), (len(rows), per_call)
    calls = [rows[i * per_call:(i + 1) * per_call] for i in range(10, 40)]

    def short(n):
        n = n.replace("(anonymous namespace)::", "").replace("rcfm::fftk::", "").replace("rcfm::narrow::fftk::", "narrow::")
        n = n.replace("rcfm::narrow::", "narrow::").replace("rcfm::", "").replace("void ", "").split("(")[0]
        return n if len(n) <= 86 else n[:83] + "..."

    print("%s.run, 240000 -> 48000, %s, %d launches per call (median of 30 calls)" %
          (kind, "synchronised after every call" if sync else "calls queued back to back", per_call))
    print("| # | kernel | device us | gap before us |")
    print("|---:|---|---:|---:|")
    busy = 0.0
    for j in range(per_call):
        dur = statistics.median((int(c[j]["End_Timestamp"]) - int(c[j]["Start_Timestamp"])) / 1e3 for c in calls)
        gap = statistics.median((int(c[j]["Start_Timestamp"]) - int(c[j - 1]["End_Timestamp"])) / 1e3 for c in calls) if j else 0.0
        busy += dur
        print("| %d | `%s` | %.2f | %.2f |" % (j + 1, short(names[10 * per_call + j]), dur, gap))
    chain = statistics.median((int(c[-1]["End_Timestamp"]) - int(c[0]["Start_Timestamp"])) / 1e3 for c in calls)
    period = statistics.median((int(calls[i + 1][0]["Start_Timestamp"]) - int(calls[i][0]["Start_Timestamp"])) / 1e3
                               for i in range(len(calls) - 1))
    print()
    print("device busy %.1f us, chain (first start .. last end) %.1f us, call period %.1f us" % (busy, chain, period))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One-off regression sweep of the batched pipeline (GPU box): random channel / audio sizes (engine lengths, lengths with
other prime factors, odd sizes, up-sampling), demodulators, channel counts and chunkings through Tuner.run_all against
the oracle loop, two buffers each.   python tools/fuzz_pipeline.py [count] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402

import radiocore as rc  # noqa: E402
import radiocore_oracle as oracle  # noqa: E402
import workloads  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-30))


def main(count=40, seed=3):
    rng = np.random.default_rng(seed)
    Bs = [24000, 25000, 30000, 32000, 36000, 40000, 48000, 50000, 60000, 64000, 75000, 80000, 96000, 100000,
          24001, 30030, 46000, 62500, 65536, 12500, 12500, 12000, 10000, 44100, 88200, 56000, 49000]      # (the last three: LDS-resident chain with A = 8000 / 6250)
    worst = 0.0
    for i in range(count):
        B = int(rng.choice(Bs))
        kind = str(rng.choice(["FM", "MFM", "WBFM"]))
        A = int(rng.choice([B // 5, B // 4, B // 2, B // 3 + 1, 8000, 8000, 6250, 12000, 16000, 9999, 11025, 22050] if kind != "WBFM" else [B // 5, B // 4, B // 2, 12000, 16000, 22050, 11025]))
        A = max(A, 300)
        if kind == "WBFM" and (B < 45000 or B % 2):
            B = 60000                                   # the 19 kHz pilot band-pass needs the rate; keep even sizes
        C = int(rng.integers(1, 6))
        chunk = int(rng.choice([0, 1, 2, 3]))
        raster = int(B * rng.choice([0.8, 1.0, 1.25]))
        centres = workloads.channel_grid(C, raster)
        tuner, ref = rc.Tuner(), oracle.Tuner()
        for f in centres:
            tuner.add_channel(f, B, getattr(rc, kind)(B, A))
            ref.add_channel(f, B, getattr(oracle, kind)(B, A))
        N = int(ref.input_bandwidth) + int(rng.choice([0, B, 2 * B]))
        tuner.request_bandwidth(float(N))
        ref.request_bandwidth(float(N))
        dev = 75e3 * B / 240000.0 if kind == "WBFM" else 0.15 * B
        x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.3, stereo=(kind == "WBFM"), deviation=dev)
        ch = 2 if kind == "WBFM" else 1
        err = 0.0
        for buf in range(2):
            xb = np.roll(x, 311 * buf)
            tuner.load(xb)
            ref.load(xb)
            audio = tuner.run_all(chunk=chunk)
            for c in ref.channels():
                want = np.asarray(c.demodulator.run(ref.run_pruned(c.index))).reshape(A, ch)
                err = max(err, rel(audio[c.index], want))
        worst = max(worst, err)
        print("%-4s B=%6d A=%6d C=%d chunk=%d N=%8d err=%.2e%s" % (kind, B, A, C, chunk, N, err, "" if err <= 1e-4 else "   <-- FAIL"),
              flush=True)
        if not err <= 1e-4:
            sys.exit(1)
    print("ok: %d cases, worst %.2e" % (count, worst))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))

#!/usr/bin/env python3
"""EVERY channel of a full-size configuration against the oracle (GPU box; minutes of host time): the HIP path's
`run_all()` audio of all C channels, `buffers` consecutive buffers (de-emphasis state), compared channel by channel with
the oracle's Tuner.run_pruned + demodulator (a pool of worker processes).  The GPU tests check a rotating sample of 64
channels per run (tests/test_hip_whole_output.py); this is the census.

    python tools/oracle_all_channels.py [cfg4|cfg3|cfg5] [buffers] [channels per batch]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    buffers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    import torch
    import radiocore as rc
    import radiocore_oracle as oracle
    import test_hip_whole_output as two
    from conftest import rel_err
    N, C, B, A, kind, x, centres, f_in = two._setup(rc, name)
    tuner = two._tuner(rc, kind, centres, B, A, N)
    ref = oracle.Tuner()
    for f in centres:
        ref.add_channel(f, B, None)
    ref.request_bandwidth(float(N))
    t0 = time.time()
    got, spectra = [], []
    x_host = x.cpu().numpy()
    for buf in range(buffers):
        if buf:
            x = torch.roll(x, 1237 * buf)
            x_host = np.roll(x_host, 1237 * buf)
        tuner.load(x)
        got.append(tuner.run_all(numpy_output=False).cpu().numpy())
        ref.load(x_host)
        spectra.append(ref._buffer.copy() if hasattr(ref, "_buffer") else None)
        print("buffer %d: HIP audio and oracle spectrum ready (%.0f s)" % (buf, time.time() - t0), flush=True)
    errs = np.zeros(C)
    for c0 in range(0, C, per):
        sample = list(range(c0, min(c0 + per, C)))
        iqs = {i: [] for i in sample}
        for buf in range(buffers):
            ref._buffer = spectra[buf]
            for i in sample:
                iqs[i].append(np.asarray(ref.run_pruned(i)))
        want = two._oracle_audio(kind, B, A, iqs)
        for i in sample:
            errs[i] = max(rel_err(got[buf][i], want[i][buf]) for buf in range(buffers))
        print("channels %4d..%4d: worst %.2e (%.0f s)" % (sample[0], sample[-1], errs[sample].max(), time.time() - t0), flush=True)
    order = np.argsort(-errs)
    print("%s: ALL %d channels x %d buffers against the oracle: worst channel %d at %.2e, median %.2e, tolerance 1e-4: %s"
          % (name, C, buffers, order[0], errs[order[0]], float(np.median(errs)), "PASS" if errs.max() <= 1e-4 else "FAIL"))
    print("ten worst:", ", ".join("%d: %.2e" % (i, errs[i]) for i in order[:10]))
    sys.exit(0 if errs.max() <= 1e-4 else 1)


if __name__ == "__main__":
    main()

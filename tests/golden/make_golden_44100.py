#!/usr/bin/env python3
"""Round-6 fixtures from the *reference* (build container only, needs /root/reference):

    python tests/golden/make_golden_44100.py   ->  tests/golden/wbfm_44100.npz

WBFM 240 000 -> 44 100 (CD audio rate: 44 100 = 2^2 3^2 5^2 7^2, the length that used to send every demodulator
transform to rocFFT and now runs through the engine's radix-7 butterfly) and the reference's own single-station
geometry 250 000 -> 48 000 (examples/receive_fm.py:18-19), WBFM and MFM, two consecutive buffers each (de-emphasis
state).  Same conventions as make_golden.py: outputs + input digests, no reference source text."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import make_golden as mg  # noqa: E402  (puts the repo root on sys.path, provides import_reference / digest / save)
import workloads  # noqa: E402

CASES = (("WBFM", 240000, 44100), ("WBFM", 250000, 48000), ("MFM", 250000, 48000), ("MFM", 240000, 44100))


def main():
    import warnings
    rc = mg.import_reference()
    warnings.filterwarnings("ignore")
    out = {}
    for kind, B, A in CASES:
        d = getattr(rc, kind)(B, A)
        for k in range(2):
            x = workloads.single_channel(B, i=6 + k, stereo=(kind == "WBFM"))
            out["%s%d_%d_%d" % (kind.lower(), k, B, A)] = d.run(x)
            out["in_%s%d_%d_%d" % (kind.lower(), k, B, A)] = np.array(mg.digest(x))
    mg.save("wbfm_44100", **out)


if __name__ == "__main__":
    main()

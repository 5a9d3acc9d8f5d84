#!/usr/bin/env python3
"""Soak (GPU box): `buffers` different wideband buffers through two lanes (two streams, shared de-emphasis state, RCFM_OPT_STATE_FENCE)
must give, buffer for buffer and bit for bit, the audio of the one-stream loop -- at full size, where the launches are long
enough to overlap in every phase.   python tools/soak_lanes.py [config] [buffers] [lanes]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from radiocore._internal import hip  # noqa: E402


def main(config="cfg3", buffers=60, lanes=2):
    lib = hip.lib()
    hip.torch()
    N, C, B, A, raster, kind = bench.CONFIGS[config]
    ch = 2 if kind == "WBFM" else 1
    kid = {"FM": 0, "MFM": 1, "WBFM": 2}[kind]
    x, centres, f_in = bench.synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    xs = [x, torch.roll(x, 1000), torch.roll(x, 12345) * 0.8, torch.roll(x, 777) * 1.1]   # four different seconds, cycled

    def make(nsets):
        sets = []
        for k in range(nsets):
            t, d = ctypes.c_void_p(), ctypes.c_void_p()
            hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
            hip.check(lib.rcfm_demod_create(kid, C, B, A, 75e-6, 0, ctypes.byref(d)))
            if k:
                hip.check(lib.rcfm_demod_bind_state(d, sets[0][1], 0, 0, hip.stream()))
            sets.append((t, d, torch.cuda.Stream()))
        if nsets > 1:
            hip.check(lib.rcfm_demod_set_option(sets[0][1], 5, 1))
        torch.cuda.synchronize()
        return sets

    def run(sets):
        outs = []
        for i in range(buffers):
            t, d, st = sets[i % len(sets)]
            s = ctypes.c_void_p(st.cuda_stream)
            audio = torch.empty((C, A, ch), dtype=torch.float32, device="cuda")
            hip.check(lib.rcfm_tuner_load(t, hip.ptr(xs[i % len(xs)]), s))
            hip.check(lib.rcfm_pipeline_run(t, d, 0, C, hip.ptr(audio), s))
            outs.append(audio)
        torch.cuda.synchronize()
        return outs

    want = run(make(1))
    got = run(make(lanes))
    bad = [i for i in range(buffers) if not torch.equal(want[i], got[i])]
    state_dep = not torch.equal(want[4], want[0])      # buffers 0 and 4 have the same input: only the carried state differs
    print("%s: %d buffers through %d lanes: %s; the state links the buffers: %s" %
          (config, buffers, lanes, "bit-identical to the one-stream loop" if not bad else "DIFFERENT at %s" % bad[:8], state_dep))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "cfg3", int(sys.argv[2]) if len(sys.argv) > 2 else 60,
                  int(sys.argv[3]) if len(sys.argv) > 3 else 2))

#!/usr/bin/env python3
"""Same-process, same-ADDRESS A/B of several builds of librcfm.so (GPU box).

    python tools/ab_libs.py [--config cfg4] [--reps 4] [--steps 10] [--stages] name=path/to/librcfm.so ...

Every build is loaded side by side (ctypes, one copy per path).  ONE device block holds the workspaces of whichever build is
being timed: its handle set is created inside an arena over that block (rcfm_arena_adopt), timed, destroyed, and the
next build's set is created over the same block -- the same allocation sequence lands on the same addresses, so the
placement lottery of profiles/r04_k_placement.md (1.5 - 4 % of a cfg4 step from where hipMalloc puts the workspaces)
is the same for every build and what is left is the code.  The input buffer is shared.  Prints per build the step
times of every repetition, their mean, and (--stages) the per-stage HIP-event times of the last repetition.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radio-core_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from radiocore._internal import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--stages", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="name:NAME=VALUE -- a demodulator option for one build")
    ap.add_argument("--tuner-opt", action="append", default=[], help="name:OPTION=VALUE -- rcfm_tuner_set_option (numeric) for one build")
    ap.add_argument("libs", nargs="+", help="name=path")
    args = ap.parse_args()
    hip.torch()
    base = hip.lib()
    libs = []
    for item in args.libs:
        name, _, path = item.partition("=")
        libs.append((name, hip.load_library(os.path.abspath(path), strict=False)))
    opts = {}
    for item in args.opt:
        name, _, rest = item.partition(":")
        opts.setdefault(name, []).append(rest)
    topts = {}
    for item in args.tuner_opt:
        name, _, rest = item.partition(":")
        topts.setdefault(name, []).append(rest)
    N, C, B, A, raster, kind = bench.CONFIGS[args.config]
    ch = 2 if kind == "WBFM" else 1
    x, centres, f_in = bench.synth_wideband_on_device(N, C, B, raster, kind, base, hip)
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    chunk_ch = min(8192, max(1024, 1024 * 240000 // B))
    want = int(17.9 * N + 64.0 * min(C, chunk_ch) * B) + (1 << 30)
    block = torch.empty(want, dtype=torch.uint8, device="cuda")
    audio = torch.empty((C, A, ch), dtype=torch.float32, device="cuda")
    s = hip.stream()
    times = {name: [] for name, _ in libs}
    stages = {}
    for rep in range(args.reps):
        for name, lib in libs:
            arena, t, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            hip.check(lib.rcfm_arena_adopt(hip.ptr(block), ctypes.c_size_t(want), ctypes.byref(arena)))
            hip.check(lib.rcfm_arena_bind(arena))
            hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
            hip.check(lib.rcfm_tuner_shard(t, 0, C))
            hip.check(lib.rcfm_demod_create({"FM": 0, "MFM": 1, "WBFM": 2}[kind], C, B, A, 75e-6, 0, ctypes.byref(d)))
            hip.check(lib.rcfm_arena_bind(None))
            for item in topts.get(name, []):
                k, _, v = item.partition("=")
                if hasattr(lib, "rcfm_tuner_set_option"):
                    hip.check(lib.rcfm_tuner_set_option(t, int(k), int(v)))
            for item in opts.get(name, []):
                k, _, v = item.partition("=")
                hip.check(lib.rcfm_demod_set_option(d, bench.DEMOD_OPTIONS[k], int(v)))

            def step():
                hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), s))
                hip.check(lib.rcfm_pipeline_run(t, d, 0, C, hip.ptr(audio), s))

            for _ in range(3):
                step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.steps):
                step()
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b) / args.steps)
            if args.stages and rep == args.reps - 1:
                lib.rcfm_profile_reset()
                lib.rcfm_profile_enable(ctypes.c_uint64((1 << lib.rcfm_profile_stage_count()) - 1))
                for _ in range(4):
                    step()
                torch.cuda.synchronize()
                stages[name] = {k: v[1] / 4 for k, v in bench.read_profile(lib).items() if v[1] > 0}
                lib.rcfm_profile_enable(ctypes.c_uint64(0))
            r, u, n = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
            hip.check(lib.rcfm_arena_stats(arena, ctypes.byref(r), ctypes.byref(u), ctypes.byref(n)))
            spilled = r.value > want
            hip.check(lib.rcfm_demod_destroy(d))
            hip.check(lib.rcfm_tuner_destroy(t))
            hip.check(lib.rcfm_arena_destroy(arena))
            if spilled and rep == 0:
                print("# %s: the block was too small (%d of %d bytes from hipMalloc): addresses differ" %
                      (name, r.value - want, r.value), flush=True)
    ref = sum(times[libs[0][0]]) / args.reps
    for name, _ in libs:
        v = times[name]
        m = sum(v) / len(v)
        print("%-14s mean %.4f ms  (%+.2f %% vs %s)  min %.4f : %s" %
              (name, m, 100 * (m / ref - 1), libs[0][0], min(v), " ".join("%.4f" % t for t in v)), flush=True)
    if stages:
        keys = sorted(stages[libs[0][0]], key=lambda k: -stages[libs[0][0]][k])
        print("%-16s" % "stage (ms)" + "".join("%12s" % n for n, _ in libs))
        for k in keys:
            print("%-16s" % k + "".join("%12.4f" % stages[n].get(k, 0.0) for n, _ in libs))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Throughput of the Tuner -> WBFM hot path on MI355X (BASELINE.json config 4).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one pass of the hot path over one 1-second wideband buffer that is
already resident in HBM: Tuner.load (FFT of N = 240 000 000 complex64 samples)
then, for this rank's share of the 1024 channels, Tuner.run + WBFM.run
(240 kHz -> 48 kHz stereo), then (N > 1) the RCCL gather of the audio to rank 0
(asynchronous on double-buffered blocks: it runs under the kernels of the next
buffer; every gather completes inside the timed region).
Channels shard across ranks; the wideband FFT cannot shard by channel and is
replicated, so total work is fixed as ranks grow: "scaling": "strong".

`--gpus N` with N > 1 and no launcher around it starts its own N ranks
(`python -m torch.distributed.run --nproc-per-node N`), one process per GPU; under
a launcher (WORLD_SIZE set) the world must equal --gpus.  An N > 1 launch times
BOTH partitionings of the wideband FFT back to back -- replicated, then the
rotating owner -- and publishes the FASTER one whose gathered audio was verified
as `value` (`config.parallelism` names it; the other keeps its own block,
`rotating` or `replicated`; a failing second leg leaves the replicated line with
the diagnosis in `rotating`).  Before them rank 0 times the same K steps alone
(`n1`, `speedup_vs_n1`) and keeps one buffer's audio, which each partitioning
must reproduce bit for bit (`self_check`).
N = 1 allocates --placement-sets complete handle sets and reports the median
one (`placement_spread` = fastest and slowest set).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (dominant stage, algorithmic bytes per launch / HIP-event time per
launch, against 8 TB/s) and, at N = 1, `cpu_baseline` (the numpy oracle timed on
this box's host cores on a bounded sample: one load + a few channels).
"""

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "radio-core_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import provenance  # noqa: E402
from workloads_device import synth_wideband_on_device  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip table (spec); 6.29e12 measured copy ceiling

CONFIGS = {
    # name: (N wideband samples, channels, B, A, raster Hz, demod)
    "cfg4": (240_000_000, 1024, 240_000, 48_000, 200_000, "WBFM"),
    "cfg3": (10_000_000, 64, 240_000, 48_000, 150_000, "MFM"),
    "cfg5": (100_000_000, 8192, 12_500, 8_000, 12_000, "FM"),
    "small": (2_400_000, 16, 240_000, 48_000, 140_000, "WBFM"),
    # geometries off the benchmark shape (reported under other_configs): the reference's own harness shape
    # (tests/benchmark.py:85, WBFM(256e3, 32e3)) and a 200 kHz-channel band, batched like cfg4
    "geo256k": (240_000_000, 1024, 256_000, 32_000, 220_000, "WBFM"),
    "geo200k": (240_000_000, 1024, 200_000, 40_000, 200_000, "WBFM"),
    # the reference's own single-station geometry (examples/receive_fm.py:18-19: 250 kHz -> 48 kHz), batched like cfg4
    "geo250k": (240_000_000, 1024, 250_000, 48_000, 220_000, "WBFM"),
    # cfg5's band with MFM instead of FM: the LDS-resident chain plus the de-emphasis launches
    "nbmfm": (100_000_000, 8192, 12_500, 8_000, 12_000, "MFM"),
}

# Algorithmic bytes of each stage per unit:
# per wideband sample for the tuner FFT, per channel for everything else.
def stage_bytes(name, N, B, A, kind):
    ch = 2 if kind == "WBFM" else 1
    table = {
        "tuner_fft_N": 16 * N,                 # read 8N, write 8N (once per buffer)
        "tuner_gather": 16 * B,                # read 8B bins, write 8B
        "tuner_ifft_B": 16 * B,
        "discriminator": 12 * B,               # read 8B, write 4B
        "pilot_stage": 16 * B,                 # read 8B, write m 4B + p 4B
        "rfft_B": 8 * B,                       # read 4B, write 4B (half spectrum)
        "hilbert_mask": 12 * B,                # read 4B, write 8B
        "ifft_B": 16 * B,
        "stereo_mix": 20 * B,                  # read z 8B + m 4B, write 8B
        "fft_B": 16 * B,
        "audio_spectrum": 8 * A * ch + 8 * A * ch / 2,
        "ifft_A": 16 * A if kind == "WBFM" else 8 * A,
        "deemphasis": 8 * A * ch,
        "deemph_state": 0,
        "dc_clip": 8 * A * ch,
        "lds_chain": 24 * B + 12 * A,          # tuner gather + IFFT_B (16B) and FM (8B + 12A) in one kernel
    }
    return float(table[name])


def path_bytes(N, C, B, A, kind):
    """Algorithmic bytes of the whole path per buffer (SURVEY.md 8d totals)."""
    per_ch = {"FM": 8 * B + 12 * A, "MFM": 8 * B + 20 * A, "WBFM": 48 * B + 40 * A}[kind]
    return 16.0 * N + C * (16.0 * B + per_ch)


def path_read_bytes(N, C, B, A, kind):
    """Read-only algorithmic bytes per buffer (the north star says "HBM-read roofline")."""
    per_ch = {"FM": 8 * B + 4 * A, "MFM": 8 * B + 8 * A, "WBFM": 28 * B + 16 * A}[kind]
    return 8.0 * N + C * (8.0 * B + per_ch)


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def read_profile(lib):
    n = lib.rcfm_profile_stage_count()
    out = {}
    for st in range(n):
        ms = ctypes.c_double()
        cnt = ctypes.c_int64()
        lib.rcfm_profile_read(st, ctypes.byref(ms), ctypes.byref(cnt))
        out[lib.rcfm_profile_stage_name(st).decode()] = (st, ms.value, cnt.value)
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


_CPU = {}


def _cpu_channel(i):
    """One channel of the reference's caller loop (multi_fm_server.py:100-106) on the oracle: worker body of the
    parallel CPU baseline (the tuner and its spectrum arrive through fork, copy-on-write)."""
    oracle, tuner, kind, B, A = _CPU["oracle"], _CPU["tuner"], _CPU["kind"], _CPU["B"], _CPU["A"]
    demod = getattr(oracle, kind)(B, A)
    t0 = time.perf_counter()
    out = demod.run(tuner.run(int(i)))        # reference-faithful O(N) roll + full-length window
    return int(i), time.perf_counter() - t0, out


def host_memory_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except (OSError, ValueError):
        pass
    return None


def fair_workers(requested, per_worker_gb=10.0):
    """Worker processes of the parallel CPU baseline: half the logical cores (one per physical core of an SMT-2
    host), capped by memory -- each worker rolls and windows the whole N-point spectrum like the reference
    (~8 GB live at N = 2.4e8) -- unless --cpu-workers asks for a number."""
    if requested > 0:
        return requested
    cores = max(1, (os.cpu_count() or 2) // 2)
    mem = host_memory_gb()
    return max(1, min(cores, int(mem // per_worker_gb))) if mem else min(cores, 8)


def cpu_baseline(x_host, f_in, centres, N, C, B, A, kind, nchan, workers):
    """The oracle (a numpy port of the reference's CPU path) timed on this box's host cores, on a bounded
    sample: one Tuner.load of the full buffer + `nchan` channels of Tuner.run + demod.
      single: one process, one thread, sequential channels like multi_fm_server.py:98-106;
              t = t_load + C * mean(t_channel);
      fair:   the per-channel part fanned out over `workers` processes (each rolls and windows the whole
              N-point spectrum per channel like the reference does, so the worker count is bounded by host
              memory, not by the core count); t = t_load + C / (channels per second with `workers` busy).
    Test infrastructure used as a reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multiprocessing as mp
    import radiocore_oracle as oracle
    tuner = oracle.Tuner()
    for f in centres:
        tuner.add_channel(f, B, None)
    tuner.request_bandwidth(float(N))
    t0 = time.perf_counter()
    tuner.load(x_host)
    t_load = time.perf_counter() - t0
    # the spectral window is built once and cached by the reference (tuner.py:155-157): not timed
    tuner._win = oracle.shifted_window("hann", N)
    _CPU.update(oracle=oracle, tuner=tuner, kind=kind, B=B, A=A)
    sample = [int(i) for i in np.linspace(0, C - 1, nchan).astype(int)]
    outputs, t_ch = {}, []
    for i in sample:
        _, dt, out = _cpu_channel(i)
        outputs[i] = out
        t_ch.append(dt)
    t_single = t_load + C * float(np.mean(t_ch))
    info = {
        "value": N / t_single / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
        "cpu": cpu_model(), "logical_cores": os.cpu_count(), "host_mem_available_GB": round(host_memory_gb() or 0.0, 1),
        "sample": "oracle Tuner.load on the full %d-sample buffer (%.1f s) + %d of %d channels of "
                  "Tuner.run+%s.run (mean %.2f s each), extrapolated t_load + C*t_channel = %.0f s per buffer"
                  % (N, t_load, nchan, C, kind, float(np.mean(t_ch)), t_single),
    }
    if workers > 1:
        try:
            per_worker = 2 if workers <= 16 else 1
            todo = [sample[j % len(sample)] for j in range(workers * per_worker)]
            with mp.get_context("fork").Pool(workers) as pool:
                t0 = time.perf_counter()
                pool.map(_cpu_channel, todo, chunksize=1)
                wall = time.perf_counter() - t0
            rate = len(todo) / wall
            t_fair = t_load + C / rate
            info["fair"] = {
                "value": N / t_fair / 1e6, "unit": "Msamples/s", "cores": workers,
                "pool": "min(logical cores // 2 = %d, MemAvailable // 10 GB = %s)" % (
                    max(1, (os.cpu_count() or 2) // 2), int((host_memory_gb() or 0) // 10)),
                "sample": "same load + %d channel runs on %d worker processes in %.1f s (%.2f channels/s), "
                          "extrapolated t_load + C/rate = %.0f s per buffer" % (len(todo), workers, wall, rate, t_fair),
            }
        except Exception as e:   # a baseline, never a reason to lose the bench line
            info["fair"] = {"value": None, "error": repr(e)[:200]}
    _CPU.clear()
    return outputs, info


def measure_surface(name, x, centres, steps, warmup, abi_seconds):
    """The same K steps through the class surface the north star names -- radiocore.tools.Tuner(cuda=True).load(x) +
    run_all(numpy_output=False) on a device tensor, one demodulator object per channel as in
    examples/multi_fm_server.py:127-133 -- instead of the raw ABI calls the headline times.  The steady state of that
    surface does no per-channel Python work (tests/test_tuner_bookkeeping.py), so the two must agree."""
    import radiocore as rc
    N, C, B, A, raster, kind = CONFIGS[name]
    t0 = time.perf_counter()
    tuner = rc.Tuner(cuda=True)
    cls = getattr(rc, kind)
    for f in centres:
        tuner.add_channel(f, B, cls(B, A, cuda=True))
    tuner.request_bandwidth(float(N))
    tuner.shard(0, C)          # like the ABI loop (rcfm_tuner_shard(0, C)): the FFT's last pass skips rows no channel reads
    setup = time.perf_counter() - t0
    for _ in range(max(warmup, 1)):
        tuner.load(x)
        audio = tuner.run_all(numpy_output=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tuner.load(x)
        audio = tuner.run_all(numpy_output=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"ms_per_step": round(dt * 1e3, 4), "steps": steps, "vs_abi": round(dt / abi_seconds, 4),
           "setup_s": round(setup, 3), "output_shape": list(audio.shape),
           "call": "Tuner(cuda=True).load(device tensor) + run_all(numpy_output=False), %d %s demodulator objects" % (C, kind)}
    del tuner, audio
    torch.cuda.empty_cache()
    return out


def stage_roofline(lib, step, N, B, A, kind, channels):
    """`roofline` block of one configuration: one step with every stage bracketed by HIP events (on the stream the
    kernels run on) finds the dominant stage; achieved = its algorithmic bytes per launch / its mean launch time."""
    lib.rcfm_profile_reset()
    lib.rcfm_profile_enable(ctypes.c_uint64((1 << lib.rcfm_profile_stage_count()) - 1))
    reps = 3
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    prof = read_profile(lib)
    lib.rcfm_profile_enable(ctypes.c_uint64(0))
    dominant = max(prof, key=lambda k: prof[k][1])
    _, ms, cnt = prof[dominant]
    per_launch_s = ms * 1e-3 / max(cnt, 1)
    launches_per_step = cnt / reps
    units = 1.0 if dominant == "tuner_fft_N" else channels / max(launches_per_step, 1)
    alg = stage_bytes(dominant, N, B, A, kind) * units
    total = sum(v[1] for v in prof.values())
    return {"bound": "hbm", "kernel": dominant, "achieved": round(alg / per_launch_s / 1e9, 1) if per_launch_s else 0.0,
            "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(alg / per_launch_s / HBM_PEAK, 4) if per_launch_s else 0.0,
            "traffic": None, "launch_us": round(per_launch_s * 1e6, 2), "launches_per_step": launches_per_step,
            "algorithmic_bytes_per_launch": alg, "share_of_step": round(ms / total, 3) if total else 0.0}


def measure_lanes(lib, hip, x, rolls, bws, N, C, B, A, kind, steps, warmup, one_lane_s, lanes=2):
    """`pipelined` block: the same K steps with consecutive buffers on `lanes` alternating streams, one handle set per
    stream, the demodulators sharing ONE de-emphasis state ordered across streams by RCFM_OPT_STATE_FENCE (what
    radiocore.tools.Lanes does through the class surface; tests/test_hip_lanes.py: bit-identical audio).  Never `value`:
    the headline stays one buffer at a time, whose kernel durations the roofline block can be read against."""
    ch = 2 if kind == "WBFM" else 1
    kind_id = {"FM": 0, "MFM": 1, "WBFM": 2}[kind]
    sets = []
    for k in range(lanes):
        tuner, demod = ctypes.c_void_p(), ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(tuner)))
        hip.check(lib.rcfm_tuner_shard(tuner, 0, C))
        hip.check(lib.rcfm_demod_create(kind_id, C, B, A, 75e-6, 0, ctypes.byref(demod)))
        if k:
            hip.check(lib.rcfm_demod_bind_state(demod, sets[0][1], 0, 0, hip.stream()))
        sets.append((tuner, demod, torch.empty((C, A, ch), dtype=torch.float32, device="cuda"), torch.cuda.Stream()))
    hip.check(lib.rcfm_demod_set_option(sets[0][1], 5, 1))     # RCFM_OPT_STATE_FENCE, after the bindings
    torch.cuda.synchronize()

    def step(i):
        tuner, demod, audio, st = sets[i % lanes]
        s = ctypes.c_void_p(st.cuda_stream)
        hip.check(lib.rcfm_tuner_load(tuner, hip.ptr(x), s))
        hip.check(lib.rcfm_pipeline_run(tuner, demod, 0, C, hip.ptr(audio), s))

    for i in range(warmup + lanes):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    finite = all(bool(torch.isfinite(a).all()) for _, _, a, _ in sets)
    for tuner, demod, _, _ in sets:
        hip.check(lib.rcfm_demod_destroy(demod))
        hip.check(lib.rcfm_tuner_destroy(tuner))
    del sets
    torch.cuda.empty_cache()
    alg = path_bytes(N, C, B, A, kind)
    return {"lanes": lanes, "ms_per_step": round(dt * 1e3, 4), "value": round(N / dt / 1e6, 1), "unit": "Msamples/s",
            "steps": steps, "path_hbm_frac": round(alg / dt / HBM_PEAK, 4), "vs_one_lane": round(dt / one_lane_s, 4),
            "output_finite": finite,
            "note": "consecutive buffers on alternating streams (radiocore.tools.Lanes / RCFM_OPT_STATE_FENCE); "
                    "parity: tests/test_hip_lanes.py"}


def measure_lane_pairs(lib, hip, x, sets, per_set_s, lo, mine, N, C, B, A, kind, steps, warmup):
    """`pipelined` block of the headline: two lanes on the SAME handle sets the one-lane headline was timed on, pair by
    pair -- set a's and set b's workspaces stay where they are, so `vs_one_lane` of a pair (its two-lane step time / the
    mean of its two one-lane step times) carries no placement draw of its own.  The block reports the median pair."""
    idx = list(range(len(sets)))
    pairs = [(idx[i], idx[(i + 1) % len(idx)]) for i in range(len(idx))] if len(idx) > 2 else [(0, 1)]
    rows = []
    for a, b in pairs:
        sa, sb = sets[a], sets[b]
        # ONE de-emphasis state per channel, ordered across the two streams by the fence (radiocore.tools.Lanes)
        hip.check(lib.rcfm_demod_bind_state(sb["demod"], sa["demod"], 0, 0, hip.stream()))
        hip.check(lib.rcfm_demod_set_option(sa["demod"], 5, 1))          # RCFM_OPT_STATE_FENCE, after the binding
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]

        def step(i):
            hs, st = (sa, sb)[i % 2], ctypes.c_void_p(streams[i % 2].cuda_stream)
            hip.check(lib.rcfm_tuner_load(hs["tuner"], hip.ptr(x), st))
            hip.check(lib.rcfm_pipeline_run(hs["tuner"], hs["demod"], lo, mine, hip.ptr(hs["audios"][0]), st))

        for i in range(warmup + 2):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        hip.check(lib.rcfm_demod_set_option(sa["demod"], 5, 0))
        one = 0.5 * (per_set_s[a] + per_set_s[b])
        rows.append({"sets": [a, b], "ms_per_step": round(dt * 1e3, 4), "one_lane_ms": round(one * 1e3, 4),
                     "vs_one_lane": round(dt / one, 4)})
    rows.sort(key=lambda r: r["vs_one_lane"])
    mid = rows[len(rows) // 2]
    dt = mid["ms_per_step"] * 1e-3
    finite = all(bool(torch.isfinite(hs["audios"][0]).all()) for hs in sets)
    return {"lanes": 2, "ms_per_step": mid["ms_per_step"], "value": round(N / dt / 1e6, 1), "unit": "Msamples/s",
            "steps": steps, "path_hbm_frac": round(path_bytes(N, C, B, A, kind) / dt / HBM_PEAK, 4),
            "vs_one_lane": mid["vs_one_lane"], "pairs": rows, "output_finite": finite,
            "note": "consecutive buffers on alternating streams (radiocore.tools.Lanes / RCFM_OPT_STATE_FENCE) over PAIRS of "
                    "the headline's own handle sets; per pair vs_one_lane = two-lane step / mean one-lane step of its two "
                    "sets (same addresses); the block is the median pair; parity: tests/test_hip_lanes.py"}


def measure_config(name, lib, hip, steps, warmup, chunk=0, with_surface=True):
    """One extra configuration on this GPU (cfg3 / cfg5; cfg4 is the headline): K timed steps of
    rcfm_tuner_load + rcfm_pipeline_run, input resident in HBM, same accounting as the headline."""
    N, C, B, A, raster, kind = CONFIGS[name]
    ch = 2 if kind == "WBFM" else 1
    x, centres, f_in = synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    tuner, demod = ctypes.c_void_p(), ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(tuner)))
    hip.check(lib.rcfm_tuner_shard(tuner, 0, C))
    hip.check(lib.rcfm_demod_create({"FM": 0, "MFM": 1, "WBFM": 2}[kind], C, B, A, 75e-6, chunk, ctypes.byref(demod)))
    audio = torch.empty((C, A, ch), dtype=torch.float32, device="cuda")

    def step():
        hip.check(lib.rcfm_tuner_load(tuner, hip.ptr(x), hip.stream()))
        hip.check(lib.rcfm_pipeline_run(tuner, demod, 0, C, hip.ptr(audio), hip.stream()))

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    finite = bool(torch.isfinite(audio).all())
    roof = stage_roofline(lib, step, N, B, A, kind, C)
    hip.check(lib.rcfm_demod_destroy(demod))
    hip.check(lib.rcfm_tuner_destroy(tuner))
    del audio
    torch.cuda.empty_cache()
    pipelined = measure_lanes(lib, hip, x, rolls, bws, N, C, B, A, kind, steps, warmup, dt)
    surface = measure_surface(name, x, centres, steps, warmup, dt) if with_surface else None
    del x
    torch.cuda.empty_cache()
    alg = path_bytes(N, C, B, A, kind)
    return {
        "surface": surface,
        "workload": "%s: %d-channel Tuner at %d MSPS -> %d x %s (%d -> %d Hz)" % (name, C, N // 1_000_000, C, kind, B, A),
        "ms_per_step": round(dt * 1e3, 4), "value": round(N / dt / 1e6, 1), "unit": "Msamples/s", "steps": steps,
        "path_algorithmic_GB": round(alg / 1e9, 3), "path_hbm_frac": round(alg / dt / HBM_PEAK, 4),
        "path_hbm_frac_read": round(path_read_bytes(N, C, B, A, kind) / dt / HBM_PEAK, 4),
        "roofline": roof,
        "pipelined": pipelined,
        "output_finite": finite,
        "parity": ("tests/test_hip_configs.py::test_%s_full_size_%s" % (name, kind.lower()) if name.startswith("cfg") else
                   "tests/test_hip_configs.py::test_run_all_narrowband_fm_geometry[MFM-12500-8000-0] (reduced band)" if name == "nbmfm" else
                   "tests/test_hip_configs.py::test_fused_chain_on_other_geometries[%s-%d-%d] (reduced band)" % (kind, B, A)),
    }


PATH_MAP_B = (240_000, 250_000, 256_000)
PATH_MAP_A = (32_000, 44_100, 48_000)


def measure_path_map(lib, hip, steps=10, warmup=2, C=64, N=16_000_000):
    """`other_configs.path_map`: where the cliffs are.  For B in {240 000, 250 000, 256 000} x A in {32 000, 44 100,
    48 000} (the reference's example and benchmark shapes, examples/receive_fm.py:18-19, tests/benchmark.py:85, and CD
    audio), 64 x WBFM behind a 16 MSPS tuner: which stages ran (librcfm's stage profile), which route that is, ms per
    buffer.  A length with a prime factor above 5 (44 100 = 2^2 3^2 5^2 7^2) sends every transform of that length -- here
    all of the demodulator's, which needs both B and A inside the engine -- to rocFFT."""
    cells = {}
    for B in PATH_MAP_B:
        raster = N // (C + 2)
        x, centres, f_in = synth_wideband_on_device(N, C, B, raster, "WBFM", lib, hip)
        rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
        bws = (ctypes.c_int32 * C)(*([B] * C))
        tuner = ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(tuner)))
        for A in PATH_MAP_A:
            demod = ctypes.c_void_p()
            hip.check(lib.rcfm_demod_create(2, C, B, A, 75e-6, 0, ctypes.byref(demod)))
            audio = torch.empty((C, A, 2), dtype=torch.float32, device="cuda")

            def step():
                hip.check(lib.rcfm_tuner_load(tuner, hip.ptr(x), hip.stream()))
                hip.check(lib.rcfm_pipeline_run(tuner, demod, 0, C, hip.ptr(audio), hip.stream()))

            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            lib.rcfm_profile_reset()
            lib.rcfm_profile_enable(ctypes.c_uint64((1 << lib.rcfm_profile_stage_count()) - 1))
            step()
            torch.cuda.synchronize()
            prof = read_profile(lib)
            lib.rcfm_profile_enable(ctypes.c_uint64(0))
            ran = {k: int(v[2]) for k, v in prof.items() if v[2] > 0}
            chan_ms = sum(v[1] for k, v in prof.items() if k != "tuner_fft_N")
            if "hilbert_mask" in ran:
                route = "rocFFT for every demodulator transform (a length outside the engine)"
            elif "ifft_A" in ran or "audio_spectrum" in ran:
                route = "engine, pilot chain fused, decimation B -> A as separate transforms"
            else:
                route = "engine, fully fused (two-transform tiles + decimating tile)"
            alg = C * (16.0 * B + 48.0 * B + 40.0 * A)
            cells["%d->%d" % (B, A)] = {
                "ms_per_step": round(dt * 1e3, 4), "channel_stages_ms": round(chan_ms, 4), "route": route,
                "launches_per_step": int(sum(ran.values())), "stages_run": ran,
                "channel_hbm_frac": round(alg / (chan_ms * 1e-3) / HBM_PEAK, 4) if chan_ms else None,
                "output_finite": bool(torch.isfinite(audio).all())}
            hip.check(lib.rcfm_demod_destroy(demod))
            del audio
        hip.check(lib.rcfm_tuner_destroy(tuner))
        del x
        torch.cuda.empty_cache()
    ref = cells["240000->48000"]["channel_stages_ms"]
    for c in cells.values():
        c["channel_stages_vs_240000->48000"] = round(c["channel_stages_ms"] / ref, 3) if ref else None
    return {"workload": "%d x WBFM behind a %d MSPS tuner, one buffer per step; channel_stages_ms = the step without the "
                        "wideband FFT (HIP-event stage times); channel_hbm_frac on 64 B + 40 A algorithmic bytes per channel"
                        % (C, N // 1_000_000),
            "cells": cells,
            "parity": "tests/test_hip_configs.py::test_path_map_cells_against_the_oracle[B-A] (reduced band, two buffers)"}


def measure_cfg1_cpu(reps=5):
    """BASELINE configs[0]: 1-channel MFM 240 kSPS -> 48 kSPS on the CPU (plumbing, no GPU).  The product has no CPU
    path: this times the oracle, like cpu_baseline does."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import radiocore_oracle as oracle
    import workloads
    x = workloads.single_channel(240_000, i=0)
    d = oracle.MFM(240_000, 48_000)
    d.run(x)
    t0 = time.perf_counter()
    for _ in range(reps):
        d.run(x)
    dt = (time.perf_counter() - t0) / reps
    return {"workload": "cfg1: 1-channel MFM 240000 -> 48000 on one host thread (oracle, kind: port)",
            "ms_per_step": round(dt * 1e3, 3), "value": round(240_000 / dt / 1e6, 3), "unit": "Msamples/s", "steps": reps,
            "parity": "tests/test_oracle_golden.py::test_mfm"}


def measure_batched_cfg2(lib, hip, steps, warmup, T=1024):
    """BASELINE configs[1] (one 240 kSPS WBFM channel) is 13.4 MB of algorithmic traffic: latency-bound as one
    call.  Its bandwidth is measured SURVEY.md section 8d's way: T consecutive buffers as T channels of one
    rcfm_demod_run (the demodulator's `batch`)."""
    B, A = 240_000, 48_000
    g = torch.Generator(device="cuda").manual_seed(5)
    ph = torch.cumsum(torch.randn(T, B, generator=g, device="cuda") * 0.3, dim=1)
    iq = torch.polar(torch.ones_like(ph), ph).to(torch.complex64).contiguous()
    del ph
    demod = ctypes.c_void_p()
    hip.check(lib.rcfm_demod_create(2, T, B, A, 75e-6, 0, ctypes.byref(demod)))
    audio = torch.empty((T, A, 2), dtype=torch.float32, device="cuda")

    def step():
        hip.check(lib.rcfm_demod_run(demod, 0, T, hip.ptr(iq), hip.ptr(audio), hip.stream()))

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    roof = stage_roofline(lib, step, 0, B, A, "WBFM", T)
    hip.check(lib.rcfm_demod_destroy(demod))
    alg = T * (48.0 * B + 40.0 * A)
    return {
        "roofline": roof,
        "workload": "cfg2 batched: %d one-second 240 kSPS buffers as %d channels of one WBFM.run (no tuner)" % (T, T),
        "ms_per_step": round(dt * 1e3, 4), "value": round(T * B / dt / 1e6, 1), "unit": "Msamples/s", "steps": steps,
        "path_algorithmic_GB": round(alg / 1e9, 3), "path_hbm_frac": round(alg / dt / HBM_PEAK, 4),
        "parity": "tests/test_hip_parity.py::test_batched_demod_matches_oracle, test_golden_wbfm",
    }


def measure_cfg2_single(reps=200):
    """BASELINE configs[1] as stated: ONE 240 kSPS WBFM channel on one GPU, one call per one-second buffer (the
    reference harness shape, tests/benchmark.py:29-31,85), device tensor in and out.  13.4 MB of algorithmic traffic
    spread over ten dependent launches: this is launch / tile latency, not bandwidth.  Two figures: `latency_us` with
    a synchronisation after every call (what a caller that needs the audio sees), `pipelined_us` with the calls
    queued back to back (what a stream of buffers costs)."""
    import radiocore as rc
    import workloads
    B, A = 240_000, 48_000
    out = {"workload": "cfg2 single: one WBFM.run per call, 240000 -> 48000, device in/out"}
    for kind in ("WBFM", "MFM", "FM"):
        d = getattr(rc, kind)(B, A, cuda=True)
        x = torch.from_numpy(workloads.single_channel(B, i=0, stereo=(kind == "WBFM"))).cuda()
        for _ in range(5):
            y = d.run(x, numpy_output=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            y = d.run(x, numpy_output=False)
            torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            y = d.run(x, numpy_output=False)
        torch.cuda.synchronize()
        pipe = (time.perf_counter() - t0) / reps
        ch = 2 if kind == "WBFM" else 1
        per = {"FM": 8 * B + 12 * A, "MFM": 8 * B + 20 * A, "WBFM": 48 * B + 40 * A}[kind]
        out[kind] = {"latency_us": round(lat * 1e6, 1), "pipelined_us": round(pipe * 1e6, 1),
                     "value": round(B / pipe / 1e6, 1), "unit": "Msamples/s",
                     "path_hbm_frac": round(per / pipe / HBM_PEAK, 5), "finite": bool(torch.isfinite(y).all()),
                     "shape": list(y.shape)}
        assert tuple(y.shape) == ((1, A, 2) if ch == 2 else (A, 1))
        del d, x, y
    out["parity"] = "tests/test_hip_parity.py::test_golden_wbfm, test_golden_mfm, test_golden_fm (240000 -> 48000)"
    return out


# rcfm_demod_set_option names (include/rcfm.h)
DEMOD_OPTIONS = {"lds_chain": 1, "fused_tiles": 2, "phase_link": 3, "narrow_tiles": 4, "pilot_chain": 6, "decim_tile": 7,
                 "lds_deemph": 8, "pilot_blocked": 9}


def apply_options(lib, hip, demod, opts):
    for item in opts:
        name, _, value = item.partition("=")
        if name not in DEMOD_OPTIONS or not value.lstrip("-").isdigit():
            die("--opt %s: expected one of %s as NAME=INTEGER" % (item, ", ".join(sorted(DEMOD_OPTIONS))))
        hip.check(lib.rcfm_demod_set_option(demod, DEMOD_OPTIONS[name], int(value)))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--chunk", type=int, default=0, help="channels per pass (0 = library default)")
    ap.add_argument("--cpu-channels", type=int, default=8, help="channels in the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="worker processes of the parallel ('fair') CPU baseline; 0 = size the pool from the host: "
                         "min(logical cores // 2, MemAvailable // 10 GB) -- each worker holds ~8 GB while it rolls "
                         "and windows the 240M-point spectrum; 1 = skip")
    ap.add_argument("--parallelism", default="both", choices=["both", "replicated", "rotating"],
                    help="N > 1: 'replicated' = every rank runs the whole wideband FFT and its own channels; 'rotating' = "
                         "rank i mod N owns buffer i (ingest + FFT) and sends each peer the spectrum bins its channels "
                         "read over xGMI (radiocore.tools.sharding.SpectrumRing); both gather the audio with RCCL.  "
                         "'both' (default) times replicated and then rotating in the same launch and publishes the faster "
                         "verified one as `value` (the other in its own block; `rotating.error` if the second leg fails); "
                         "at N = 1 there is nothing to partition")
    ap.add_argument("--placement-sets", type=int, default=4,
                    help="N = 1: complete handle sets (tuner + demodulator + audio block) allocated side by side; each is "
                         "timed over exactly K steps, `value` is the MEDIAN set and `placement_spread` the fastest and "
                         "slowest (where hipMalloc puts the workspaces moves a cfg4 step by 1.5-4 %: "
                         "profiles/r04_k_placement.md)")
    ap.add_argument("--arena", type=int, default=-1,
                    help="1: every handle set takes its workspaces from ONE device block (rcfm_arena_*), 0: from "
                         "hipMalloc one by one; -1 = the default of this configuration")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="rcfm_demod_set_option on every demodulator handle of the timed legs (A/B runs: tools/ab_args.sh): "
                         + ", ".join(sorted(DEMOD_OPTIONS)))
    ap.add_argument("--self-check", action="store_true", help=argparse.SUPPRESS)     # the default for N > 1
    ap.add_argument("--no-self-check", action="store_true",
                    help="N > 1: skip the one-rank leg on rank 0 (`n1`, `speedup_vs_n1`) and the bit-for-bit comparison of "
                         "one buffer through each partitioning with the single-GPU audio (`self_check`)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the other GPU configurations (cfg3, cfg5, batched cfg2) reported beside the headline")
    ap.add_argument("--profile-all", action="store_true", help="print the per-stage table to stderr as well")
    ap.add_argument("--no-pcie", action="store_true",
                    help="skip the host-fed pass (page-locked host buffer, H2D of buffer i+1 on a copy stream overlapped "
                         "with the kernels of buffer i: extra field pcie_inclusive, never `value`)")
    ap.add_argument("--pcie", action="store_true", help=argparse.SUPPRESS)     # rounds 1-4 spelling: now the default
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def die(what, code=2, **more):
    """A run that cannot measure what it was asked for ends with ONE JSON error line on stderr and a non-zero
    status -- never with a contract line that describes something else."""
    sys.stderr.write(json.dumps(dict({"error": what}, **more)) + "\n")
    sys.stderr.flush()
    sys.exit(code)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (how the driver starts N = 1): become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <the same arguments>` -- one process per GPU, rank 0 prints the one line."""
    have = torch.cuda.device_count()
    if "RCFM_BENCH_DEVICE" not in os.environ and have < args.gpus:
        die("bench.py --gpus %d: this node shows %d GPU(s)" % (args.gpus, have), gpus=args.gpus, visible=have)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


class Run:
    """What every leg of one bench.py process shares: the rank's place in the world, the process group, the watchdog,
    the library and the resident wideband buffer."""

    def __init__(self, args):
        self.args = args
        self.rank, self.world, local = dist_env()
        # Dry run of the N > 1 code on a one-GPU box (tests/test_bench_multirank.py): RCFM_BENCH_DEVICE puts every rank
        # on that device and RCFM_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU); the audio blocks
        # then travel through host memory.  The numbers of such a run mean nothing.
        self.backend = os.environ.get("RCFM_BENCH_BACKEND", "nccl")
        self.local = int(os.environ.get("RCFM_BENCH_DEVICE", local))
        # RCFM_BENCH_FORCE_DIST=1 (tests/test_nccl_world1.py): a ONE-rank launch goes down the N > 1 code path -- RCCL
        # process group, asynchronous double-buffered gather into views of the result, barrier, max-over-ranks -- which
        # a one-GPU box can execute (a one-rank communicator is legal) although it cannot execute N > 1 itself.
        self.multi = self.world > 1 or os.environ.get("RCFM_BENCH_FORCE_DIST") == "1"
        self.dist = None
        self.rccl_ranks = None
        # N > 1 only: a rank that stops making progress (a peer died, a transfer never matched) must end the run with a
        # message instead of holding the node until the lease expires.  `phase` says where it was.
        self.phase = {"name": "init", "step": -1, "t": time.monotonic(), "leg": "setup"}
        self.limit = float(os.environ.get("RCFM_BENCH_TIMEOUT", "300"))
        # the line of the legs that have finished: a later leg that fails still publishes them (rank 0)
        self.finished = None
        # --self-check (default for N > 1): the one-rank leg + the bit-for-bit comparison of both partitionings with it
        self.self_check = self.multi and not args.no_self_check

    def progress(self, name, step=-1):
        self.phase.update(name=name, step=step, t=time.monotonic())

    def fail(self, what):
        """End of the run from any thread.  While the SECOND partitioning of a 'both' launch runs, the first one's line
        is already measured: rank 0 prints it with the failure in the `rotating` block and every rank leaves with
        status 0 -- a first hardware run yields a number and a diagnosis, not a hung lease."""
        ph = self.phase
        msg = {"error": what, "rank": self.rank, "world": self.world, "phase": ph["name"], "step": ph["step"],
               "parallelism": ph["leg"], "backend": self.backend}
        sys.stderr.write(json.dumps(msg) + "\n")
        sys.stderr.flush()
        if ph["leg"] == "rotating (second leg)":
            if self.rank == 0 and self.finished is not None:
                self.finished["rotating"] = msg
                sys.stdout.write(json.dumps(self.finished) + "\n")
                sys.stdout.flush()
            os._exit(0)
        os._exit(3)

    def start_group(self):
        import datetime
        import threading
        import torch.distributed as dist
        self.dist = dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("RCFM_BENCH_FORCE_DIST") == "1":
            os.environ.setdefault("RCFM_GATHER_FORCE_COLLECTIVE", "1")   # exercise the collective on a one-rank group

        def watchdog():
            while True:
                time.sleep(1.0)
                idle = time.monotonic() - self.phase["t"]
                if self.phase["name"] == "done":
                    return
                if idle > self.limit:
                    self.fail("bench.py made no progress for %.0f s" % idle)

        threading.Thread(target=watchdog, daemon=True).start()
        to = datetime.timedelta(seconds=self.limit)
        self.progress("process group rendezvous")
        try:
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local), timeout=to)
            else:
                dist.init_process_group(self.backend, timeout=to)
        except Exception as e:                                   # a peer never arrived
            self.fail("bench.py made no progress: process group rendezvous failed (%s: %s)" % (type(e).__name__, str(e)[:200]))
        self.progress("process group up")
        if self.backend == "nccl":
            # the communicator exists once a collective has run on it: its size is what RCCL saw
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)
            self.rccl_ranks = int(probe.item())
        else:
            self.rccl_ranks = 0          # dry run: no RCCL communicator at all

    def max_over_ranks(self, seconds):
        if not self.multi:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def fft_launch_names(lib, hip, N):
    """The launches behind the stage `tuner_fft_N`: one k_fft_tile pass per factor of the forward plan."""
    plan = hip.FftPlan()
    if lib.rcfm_fft_describe(ctypes.c_int64(N), 0, ctypes.byref(plan)) != 0:
        return ["rocFFT (length outside the engine)"]
    return ["k_fft_tile<L=%d> (pass %d of %d)" % (plan.passes[i].L,
                                                  i + 1, plan.npass) for i in range(plan.npass)]


def stage_table(prof, N, B, A, kind, channels, steps=1):
    """`stages` block: per stage of ONE step the HIP-event milliseconds, the number of launches, and the algorithmic
    bytes those launches account for (stage_bytes x units) -- every roofline figure of the line can be recomputed
    from this block: frac = bytes / (ms * 1e-3) / 8e12."""
    out = {}
    for k, (st, ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        if cnt == 0:
            continue
        by = stage_bytes(k, N, B, A, kind) * (1 if k == "tuner_fft_N" else channels)
        out[k] = {"ms": round(ms / steps, 4), "launches": cnt / steps, "algorithmic_bytes": by,
                  "TBps": round(by / (ms / steps * 1e-3) / 1e12, 3) if ms else 0.0}
    return out


def measure_partitioning(run, x, centres, f_in, parallelism, primary, ref=None):
    """One leg: K timed steps of the hot path under one partitioning of the work over the ranks.  `primary`: the leg
    the line's `value` comes from (N = 1: also cpu_baseline, parity spot check)."""
    args, rank, world, multi, backend = run.args, run.rank, run.world, run.multi, run.backend
    dist = run.dist
    progress = run.progress
    from radiocore._internal import hip
    from radiocore.tools import sharding
    lib = hip.lib()
    N, C, B, A, raster, kind = CONFIGS[args.config]
    ch = 2 if kind == "WBFM" else 1
    rotating = parallelism == "rotating"

    # this rank's contiguous share of the channels
    lo, hi = sharding.channel_range(rank, world, C)
    mine = hi - lo
    rolls = [int(f_in - f) for f in centres]
    roll_a = (ctypes.c_int64 * C)(*rolls)
    bw_a = (ctypes.c_int32 * C)(*([B] * C))
    kind_id = {"FM": 0, "MFM": 1, "WBFM": 2}[kind]
    # N > 1: the audio blocks are double-buffered so that the gather of buffer i (RCCL's own stream, xGMI)
    # overlaps the kernels of buffer i+1; every gather completes inside the timed region (barrier()).
    nbuf = 2 if multi else 1
    # N = 1: several complete handle sets side by side (their workspaces land on different memory: placement_spread)
    nsets = max(1, args.placement_sets) if (not multi and not rotating) else 1
    use_arena = bool(args.arena) if args.arena >= 0 else False
    sets = []
    for k in range(nsets):
        arena = ctypes.c_void_p()
        if use_arena:
            # one block for the whole set: the tuner's spectrum + scratch (16 N bytes) and the per-chunk workspaces
            # (at most ~64 bytes per channel sample of a chunk); whatever does not fit goes to a second block
            chunk_ch = args.chunk if args.chunk > 0 else min(8192, max(1024, 1024 * 240000 // B))
            want = int(17.6 * N + 64.0 * min(mine, chunk_ch) * B) + (1 << 30)
            hip.check(lib.rcfm_arena_create(ctypes.c_size_t(want), ctypes.byref(arena)))
            hip.check(lib.rcfm_arena_bind(arena))
        tuner, demod = ctypes.c_void_p(), ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, roll_a, bw_a, ctypes.byref(tuner)))
        hip.check(lib.rcfm_tuner_shard(tuner, lo, mine))   # the wideband FFT keeps only what this rank's channels read
        hip.check(lib.rcfm_demod_create(kind_id, C, B, A, 75e-6, args.chunk, ctypes.byref(demod)))
        apply_options(lib, hip, demod, args.opt)
        if use_arena:
            hip.check(lib.rcfm_arena_bind(None))
        sets.append({"tuner": tuner, "demod": demod, "arena": arena,
                     "audios": [torch.empty((mine, A, ch), dtype=torch.float32, device="cuda") for _ in range(nbuf)]})
    cur = {"set": sets[0]}
    tuner, demod = sets[0]["tuner"], sets[0]["demod"]
    ring = surf = None
    if rotating:
        # the rotating owner runs through the class surface (Tuner + one demodulator object per channel): the ring
        # swaps the tuner's spectrum storage between buffers, which the surface exposes (attach / window / adopt)
        import radiocore as rc
        surf = rc.Tuner(cuda=True)
        for f in centres:
            surf.add_channel(f, B, getattr(rc, kind)(B, A, cuda=True))
        surf.request_bandwidth(float(N))
        ring = sharding.SpectrumRing(surf, N, C)
        for j in range(ring.lookahead):                       # prime: `lookahead` buffers in flight from here on
            ring.submit(j, x if ring.owner(j) == rank else None)
    gathereds = [torch.empty((C, A, ch), dtype=torch.float32, device="cuda") if (multi and rank == 0) else None
                 for _ in range(nbuf)]
    in_flight = [None] * nbuf
    counter = [0]
    source = [x]               # what an owner ingests: the resident buffer, or (pcie_inclusive) page-locked host memory

    def step():
        s = hip.stream()
        slot = counter[0] % nbuf
        counter[0] += 1
        hs = cur["set"]
        if in_flight[slot] is not None:
            in_flight[slot].wait()          # stream-ordered: this slot's previous block has left
            in_flight[slot] = None
        if rotating:
            i = counter[0] - 1
            j = i + ring.lookahead
            ring.submit(j, source[0] if ring.owner(j) == rank else None)   # owner of buffer j: ingest + FFT + sends
            ring.acquire(i)                                                # buffer i's bins for this rank's channels
            hs["audios"][slot] = surf.run_all(numpy_output=False)
        else:
            hip.check(lib.rcfm_tuner_load(hs["tuner"], hip.ptr(x), s))
            # pipeline_run addresses channels of tuner and demod by the same index
            hip.check(lib.rcfm_pipeline_run(hs["tuner"], hs["demod"], lo, mine, hip.ptr(hs["audios"][slot]), s))
        if multi and backend == "nccl":     # RCCL over xGMI: the only collective on the path
            in_flight[slot] = sharding.gather_audio(hs["audios"][slot], C, dst=0, out=gathereds[slot], async_op=True)
        elif multi:                         # dry run: same protocol through host memory
            got = sharding.gather_audio(hs["audios"][slot].cpu(), C, dst=0)
            if rank == 0:
                gathereds[slot].copy_(got)

    def barrier():
        for i in range(nbuf):
            if in_flight[i] is not None:
                in_flight[i].wait()
                in_flight[i] = None
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def profile_pass(steps_):
        """Untimed: `steps_` steps with every stage bracketed by HIP events on the stream the kernels run on."""
        lib.rcfm_profile_reset()
        lib.rcfm_profile_enable(ctypes.c_uint64((1 << lib.rcfm_profile_stage_count()) - 1))
        for _ in range(steps_):
            step()
        torch.cuda.synchronize()
        prof = {k: (st, ms / steps_, cnt / steps_) for k, (st, ms, cnt) in read_profile(lib).items()}
        lib.rcfm_profile_enable(ctypes.c_uint64(0))
        return prof

    for hs in sets:
        cur["set"] = hs
        for k in range(args.warmup):
            progress("warm-up", k)
            step()
    progress("warm-up barrier")
    barrier()

    # pass 1 (untimed): every stage bracketed, to find the dominant one
    cur["set"] = sets[0]
    prof_steps = world if rotating else 1      # a rank runs the wideband FFT once per `world` buffers when it rotates
    prof_all = profile_pass(prof_steps)
    dominant = max(prof_all, key=lambda k: prof_all[k][1])

    # timed region: exactly K steps, barrier + synchronize on both sides; only the dominant stage keeps its event
    # pairs (on the stream the kernels run on).  N = 1: once per handle set, the MEDIAN set is the line's value.
    timed = []
    for hs in sets:
        cur["set"] = hs
        lib.rcfm_profile_reset()
        lib.rcfm_profile_enable(ctypes.c_uint64(1 << prof_all[dominant][0]))
        progress("barrier before the timed region")
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            progress("timed step", k)
            step()
        progress("barrier after the timed region")
        barrier()
        dt = time.perf_counter() - t0
        timed.append((dt, read_profile(lib)[dominant], hs))
        lib.rcfm_profile_enable(ctypes.c_uint64(0))
    order = sorted(range(len(timed)), key=lambda i: timed[i][0])
    pick = order[len(order) // 2]                 # 4 sets: the third fastest (the upper median)
    elapsed, dom, median_set = timed[pick]
    own_elapsed = elapsed
    cur["set"] = median_set
    tuner, demod = median_set["tuner"], median_set["demod"]
    audio = median_set["audios"][0]
    if nsets > 1:                                 # the decomposition of the set the value comes from
        prof_all = profile_pass(1)

    lanes_block = None
    if primary and not multi and nsets >= 2 and not args.no_extras:
        progress("two lanes over pairs of the handle sets")
        lanes_block = measure_lane_pairs(lib, hip, x, sets, [t[0] / args.steps for t in timed], lo, mine, N, C, B, A, kind,
                                         args.steps, args.warmup)

    per_rank = None
    if multi:
        progress("max over ranks")
        elapsed = run.max_over_ranks(elapsed)
        # where each rank's step went (HIP-event stage times of the untimed profile pass; the ring's own event pairs,
        # collected in a pass of their own AFTER the timed region: the timed steps carry no instrumentation):
        # the first thing to read when a multi-GPU number looks wrong
        fft_pb = prof_all["tuner_fft_N"][1]                       # per buffer of this rank (rotating: one in `world`)
        chan_pb = sum(v[1] for k, v in prof_all.items() if k != "tuner_fft_N")
        mine_ms = 1e3 * own_elapsed / args.steps
        row = {"rank": rank, "channels": mine, "step_ms": round(mine_ms, 4), "fft_ms": round(fft_pb, 4),
               "chan_ms": round(chan_pb, 4), "send_ms": 0.0, "wait_ms": 0.0}
        if ring is not None:
            progress("ring timing pass")
            ring.enable_timing()
            for _ in range(2 * world):
                step()
            barrier()
            tsum = ring.timing_summary() or {}
            ring.disable_timing()
            row.update(fft_ms=tsum.get("fft_ms", 0.0), send_ms=tsum.get("send_ms", 0.0), wait_ms=tsum.get("wait_ms", 0.0),
                       owned_buffers=tsum.get("fft_count", 0))
        row["other_ms"] = round(mine_ms - row["chan_ms"] - row["wait_ms"] - (0.0 if rotating else row["fft_ms"]), 4)
        progress("per-rank rows")
        rows = [None] * world
        dist.all_gather_object(rows, row)
        per_rank = rows

    ms_per_step = 1e3 * elapsed / args.steps
    value = N * args.steps / elapsed / 1e6
    launches_per_step = dom[2] / args.steps
    per_launch_s = dom[1] * 1e-3 / max(dom[2], 1)
    units = 1.0 if dominant == "tuner_fft_N" else mine / max(launches_per_step, 1)
    alg_bytes = stage_bytes(dominant, N, B, A, kind) * units
    achieved = alg_bytes / per_launch_s if per_launch_s else 0.0
    total_alg = 16.0 * N + mine * (path_bytes(N, C, B, A, kind) - 16.0 * N) / C
    total_read = 8.0 * N + mine * (path_read_bytes(N, C, B, A, kind) - 8.0 * N) / C

    # HBM traffic of the dominant stage from the committed rocprofv3 PMC passes (tools/profile_traffic.sh ->
    # tools/traffic_summary.py -> profiles/hbm_traffic.json).  The table is stamped with a digest of the kernel
    # sources it was collected on (provenance.py): counters of other device code are withheld, not quoted.
    traffic, traffic_source, traffic_stale = None, None, False
    # the committed PMC passes describe cfg4 with the default chunking on one GPU: nothing else is claimed
    if args.config == "cfg4" and world == 1 and args.chunk == 0:
        traffic, traffic_source, traffic_stale = provenance.stage_traffic(dominant)

    launches = fft_launch_names(lib, hip, N) if dominant == "tuner_fft_N" else [dominant]
    result = {
        "metric": "IQ Msamples/s through Tuner+%s at %d channels" % (kind, C),
        "value": round(value, 2),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s: %d-channel Tuner at %d MSPS complex64 -> %d x %s (%d -> %d Hz), 1-second buffers, "
                        "input resident in HBM" % (args.config, C, N // 1_000_000, C, kind, B, A),
            "channels": C, "channels_per_gpu": mine, "wideband_samples": N, "channel_samples": B,
            "audio_samples": A, "parallelism": (
                "channels sharded x%d, rotating FFT owner (rank i mod %d ingests and transforms buffer i, peers receive "
                "their spectrum windows over xGMI, %d buffers in flight), RCCL gather" % (world, world, ring.lookahead)
                if rotating else "channels sharded x%d, wideband FFT replicated, RCCL gather" % world
                if world > 1 else "single GPU"),
        },
        "path_hbm_frac": round(total_alg / (ms_per_step * 1e-3) / HBM_PEAK, 4),
        "path_hbm_frac_read": round(total_read / (ms_per_step * 1e-3) / HBM_PEAK, 4),
        "path_algorithmic_GB": round(total_alg / 1e9, 3),
        "path_algorithmic_read_GB": round(total_read / 1e9, 3),
        "kernel_source_sha": provenance.kernel_source_sha(),
        "roofline": {
            "bound": "hbm", "kernel": " + ".join(launches), "stage": dominant,
            "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic, "traffic_stale": traffic_stale,
            "traffic_source": traffic_source,
            "launch_us": round(per_launch_s * 1e6, 2), "launches_per_step": launches_per_step,
            "kernel_launches_per_stage_launch": len(launches),
            "algorithmic_bytes_per_launch": alg_bytes,
            "share_of_step": round(dom[1] / args.steps / ms_per_step, 3),
            "note": "a stage launch = the %d kernel launches named in `kernel`, bracketed by one HIP-event pair on their "
                    "stream inside the timed region; achieved = algorithmic_bytes_per_launch / launch_us" % len(launches),
        },
        # one step of the set `value` comes from, every stage bracketed (untimed pass after the timed region)
        "stages": stage_table(prof_all, N, B, A, kind, mine),
    }
    if nsets > 1:
        per_set = [round(1e3 * t[0] / args.steps, 4) for t in timed]
        result["placement_spread"] = [min(per_set), max(per_set)]
        result["placement"] = {
            "handle_sets": nsets, "ms_per_step_by_set": per_set, "value_from_set": pick, "arena": use_arena,
            "note": "%d complete handle sets side by side, each timed over exactly %d steps between barriers; `value` / "
                    "`ms_per_step` / `roofline` / `stages` are the set with the median time (upper median of an even "
                    "count)" % (nsets, args.steps)}
    if lanes_block is not None:
        result["pipelined"] = lanes_block
    if multi:
        result["rccl_ranks"] = run.rccl_ranks

    if multi:
        # Channel sharding scales the per-channel stages; the replicated wideband FFT does not.
        # Per-channel-stage rate of this run = channel samples through (step time - this rank's FFT time), and the
        # Amdahl bound of the end-to-end speed-up at this world size from the single-GPU shares.
        fft_ms = prof_all["tuner_fft_N"][1]
        chan_ms = max(ms_per_step - fft_ms, 1e-6)
        result["channel_stage_value"] = {
            "value": round(C * B / (chan_ms * 1e-3) / 1e6, 1), "unit": "channel Msamples/s",
            "note": "all %d channels' samples / (step time - this rank's wideband-FFT time per buffer, %.3f ms on rank 0%s)"
                    % (C, fft_ms, ": one transform every %d buffers" % world if rotating else ", replicated")}
        alg_fft, alg_all = 16.0 * N, path_bytes(N, C, B, A, kind)
        # replicated FFT: Amdahl; rotating owner: every stage divides by the world size
        result["amdahl_bound_speedup"] = float(world) if rotating else round(
            alg_all / (alg_fft + (alg_all - alg_fft) / world), 3)
        # per rank: step_ms = its own K steps / K; fft_ms = one wideband FFT (replicated: every buffer; rotating: on the
        # owner's stream, once per `world` buffers); chan_ms = its channels' stages per buffer; send_ms = the owner's
        # sends of one buffer (rotating); wait_ms = how long the channel stream waited for a buffer's bins (rotating);
        # other_ms = what is left of the step (gather exposure, launch gaps)
        result["per_rank"] = per_rank
        if rotating:
            result["rotating_owner"] = {
                "lookahead": ring.lookahead, "spectrum_slots": len(ring.slots),
                "full_slots": ring.full_slots, "window_slots": ring.window_slots,
                "slot_bytes": int(ring.slot_bytes()),
                "window_bins_of_rank0": int(sum(b - a for a, b in ring.segments[0])),
                "bytes_sent_per_owned_buffer": int(ring.bytes_sent_per_buffer()),
                "ffts_per_rank_per_buffer": round(1.0 / world, 4)}

        # the last gathered block (outside the timed region): every rank's rows arrived
        progress("gather check")
        step()
        barrier()
        if rank == 0:
            g = gathereds[(counter[0] - 1) % nbuf]
            bounds = [sharding.channel_range(r, world, C) for r in range(world)]
            result["gather_check"] = {
                "finite": bool(torch.isfinite(g).all().item()),
                "blocks_with_audio": int(sum(bool((g[a:b].abs().amax() > 1e-3).item()) for a, b in bounds if b > a)),
                "blocks": int(sum(1 for a, b in bounds if b > a)),
                "own_block_equal": bool(torch.equal(g[lo:hi], cur["set"]["audios"][(counter[0] - 1) % nbuf])),
            }

    if multi and run.self_check:
        # One buffer from first-buffer state through this partitioning, gathered, against the one-rank audio rank 0
        # computed in this launch (measure_one_rank): bit for bit -- sharding only changes WHO computes a channel.
        progress("self check")
        if rotating:
            surf.reset_states()
        else:
            hip.check(lib.rcfm_demod_reset_state(cur["set"]["demod"], hip.stream()))
        step()
        barrier()
        if rank == 0:
            g = gathereds[(counter[0] - 1) % nbuf]
            if ref is None:
                result["self_check"] = {"bit_identical": False, "error": "no one-rank reference in this launch"}
            else:
                same = bool(torch.equal(g, ref))
                peak = float(ref.abs().amax().item())
                result["self_check"] = {
                    "bit_identical": same, "max_abs_diff": 0.0 if same else float((g - ref).abs().amax().item()),
                    "peak": peak, "channels": C,
                    "against": "one buffer from first-buffer state on ONE GPU (rank 0, same launch), all %d channels" % C}

    if not args.no_pcie:
        # Host-fed variant (DESIGN.md sections 4, 5): the wideband buffer starts in page-locked host memory
        # (radiocore.tools.Buffer(cuda=True)) and crosses PCIe inside the loop.
        #   replicated: EVERY rank copies the whole buffer over its own link (radiocore.tools.Feeder: two device
        #               slots, buffer i+1 copied on its own stream under the kernels of buffer i);
        #   rotating:   only the owner of a buffer copies it (SpectrumRing stages it on its FFT stream), so each link
        #               carries 1/N of the buffers.
        progress("host-fed pass")
        from radiocore.tools import Buffer, Feeder
        host = Buffer(N, dtype=np.complex64, cuda=True)
        host.data[:] = x.cpu().numpy()
        k = max(min(args.steps, 10), 4)
        if rotating:
            source[0] = host._owner.view(torch.complex64)      # the same pinned pages as a torch tensor
            for _ in range(ring.lookahead + 1):                 # the buffers already in flight came from HBM
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            barrier()
            dt = (time.perf_counter() - t0) / k
            source[0] = x
            note = "rotating owner: buffer i crosses PCIe on rank i mod %d only, on that rank's FFT stream" % world
        else:
            feeder = Feeder(N, dtype=np.complex64, depth=2)

            def consume():
                with feeder.next() as xd:
                    hip.check(lib.rcfm_tuner_load(tuner, hip.ptr(xd), hip.stream()))
                    hip.check(lib.rcfm_pipeline_run(tuner, demod, lo, mine, hip.ptr(audio), hip.stream()))

            feeder.submit(host.data)
            consume()
            barrier()
            t0 = time.perf_counter()
            feeder.submit(host.data)
            for i in range(k):
                if i + 1 < k:
                    feeder.submit(host.data)
                consume()
            barrier()
            dt = (time.perf_counter() - t0) / k
            note = ("radiocore.tools.Feeder: page-locked host buffer, H2D of buffer i+1 overlapped with the kernels of "
                    "buffer i" + ("; every rank copies the whole buffer over its own link" if world > 1 else ""))
        dt = run.max_over_ranks(dt)
        result["pcie_inclusive"] = {"ms_per_step": round(dt * 1e3, 3), "value": round(N / dt / 1e6, 1),
                                    "unit": "Msamples/s", "steps": k, "h2d_GBps": None, "note": note}
        if not rotating:
            t0 = time.perf_counter()
            feeder.submit(host.data)
            with feeder.next():
                pass
            torch.cuda.synchronize()
            result["pcie_inclusive"]["h2d_GBps"] = round(N * 8 / (time.perf_counter() - t0) / 1e9, 1)
            feeder.close()
        del host

    if primary and rank == 0 and world == 1 and args.cpu_channels > 0 and not rotating and not multi:
        progress("cpu baseline")
        x_host = x.cpu().numpy()
        ref_audio, result["cpu_baseline"] = cpu_baseline(x_host, f_in, centres, N, C, B, A, kind,
                                                         args.cpu_channels, fair_workers(args.cpu_workers))
        del x_host
        # full-size parity spot check (outside the timed region): first-buffer state on
        # both sides, the oracle's sampled channels against the GPU's
        hip.check(lib.rcfm_demod_reset_state(demod, hip.stream()))
        step()
        torch.cuda.synchronize()
        got = cur["set"]["audios"][(counter[0] - 1) % nbuf].cpu().numpy()
        worst = 0.0
        for i, want in ref_audio.items():
            want = np.asarray(want).reshape(A, ch)
            worst = max(worst, float(np.max(np.abs(got[i] - want)) / np.max(np.abs(want))))
        result["parity_vs_oracle"] = {"channels": sorted(ref_audio), "max_rel_err": worst, "tol": 1e-4}
    elif primary and rank == 0:
        result["cpu_baseline"] = None

    progress("leg teardown")
    if rotating:
        ring.drain()               # the buffers still in flight: every posted transfer completes before the group goes
        torch.cuda.synchronize()
        ring.close()
        del ring, surf
    barrier()
    for hs in sets:
        hip.check(lib.rcfm_demod_destroy(hs["demod"]))
        hip.check(lib.rcfm_tuner_destroy(hs["tuner"]))
        if hs["arena"]:
            hip.check(lib.rcfm_arena_destroy(hs["arena"]))
    del sets, gathereds, cur, median_set, audio, timed
    torch.cuda.empty_cache()
    return result, (roll_a, bw_a)


def measure_one_rank(run, x, centres, f_in):
    """N > 1 launches only, rank 0, before any data-path collective: the SAME K steps on ONE GPU (all channels, one
    handle set) -- the denominator of `speedup_vs_n1` measured in the same launch on the same box -- and the audio of
    one buffer from first-buffer state, which both partitionings must reproduce bit for bit (`self_check`).  Two
    references: the handle's default wideband plan (what the replicated partitioning runs) and the default ORDER of the
    plan (RCFM_TUNER_OPT_ALIGNED_PLAN = 0: what a rotating owner runs, whose spectrum lives in an attached slot)."""
    args = run.args
    from radiocore._internal import hip
    lib = hip.lib()
    N, C, B, A, raster, kind = CONFIGS[args.config]
    ch = 2 if kind == "WBFM" else 1
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    tuner, demod = ctypes.c_void_p(), ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(tuner)))
    hip.check(lib.rcfm_tuner_shard(tuner, 0, C))
    hip.check(lib.rcfm_demod_create({"FM": 0, "MFM": 1, "WBFM": 2}[kind], C, B, A, 75e-6, args.chunk, ctypes.byref(demod)))
    apply_options(lib, hip, demod, args.opt)
    audio = torch.empty((C, A, ch), dtype=torch.float32, device="cuda")

    def step():
        hip.check(lib.rcfm_tuner_load(tuner, hip.ptr(x), hip.stream()))
        hip.check(lib.rcfm_pipeline_run(tuner, demod, 0, C, hip.ptr(audio), hip.stream()))

    for k in range(args.warmup):
        run.progress("one-rank leg: warm-up", k)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        run.progress("one-rank leg: timed step", k)
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    refs = {}
    for name, aligned in (("replicated", 1), ("rotating", 0)):
        hip.check(lib.rcfm_tuner_set_option(tuner, hip.RCFM_TUNER_OPT_ALIGNED_PLAN, aligned))
        hip.check(lib.rcfm_demod_reset_state(demod, hip.stream()))
        step()
        torch.cuda.synchronize()
        refs[name] = audio.clone()
    hip.check(lib.rcfm_demod_destroy(demod))
    hip.check(lib.rcfm_tuner_destroy(tuner))
    del audio
    torch.cuda.empty_cache()
    return {"ms_per_step": round(dt * 1e3, 4), "value": round(N / dt / 1e6, 2), "unit": "Msamples/s", "steps": args.steps,
            "note": "rank 0 alone, all %d channels, one handle set, timed in this launch before the first data-path "
                    "collective (the other ranks wait at a barrier)" % C}, refs


def partitioning_verified(r):
    """A leg's number may become `value` only when its gathered audio was checked: every rank's block arrived with audio
    in it, rank 0's own block is the one it computed, and (when the launch had a one-rank reference) one buffer from
    first-buffer state equals the single-GPU audio bit for bit."""
    g, sc = r.get("gather_check"), r.get("self_check")
    ok = bool(g) and g["finite"] and g["own_block_equal"] and g["blocks_with_audio"] == g["blocks"]
    return ok and (sc is None or sc["bit_identical"])


def publish_faster(first, second, n1):
    """Both partitionings ran in one launch: the line's `value` is the FASTER one whose audio was verified
    (config.parallelism names it); the other one keeps its own block."""
    legs = {"replicated": first, "rotating": second}
    ok = {k: partitioning_verified(v) for k, v in legs.items()}
    order = sorted(legs, key=lambda k: -legs[k]["value"])
    winner = next((k for k in order if ok[k]), "replicated")
    loser = "rotating" if winner == "replicated" else "replicated"
    line = dict(legs[winner])
    for k in ("rotating", "replicated"):
        line.pop(k, None)
    block = {k: legs[loser][k] for k in ROTATING_KEYS if k in legs[loser]}
    block["parallelism"] = legs[loser]["config"]["parallelism"]
    block["vs_published"] = round(legs[loser]["value"] / line["value"], 4) if line["value"] else None
    line[loser] = block
    # what the first leg carried for the whole launch stays with the line
    for k in ("cpu_baseline", "rccl_ranks", "kernel_source_sha", "n1"):
        if k in first:
            line[k] = first[k]
    line["speedup_vs_n1"] = round(line["value"] / n1["value"], 4) if n1 and n1["value"] else None
    line["partitionings"] = {
        "published": winner, "verified": ok, "value": {k: legs[k]["value"] for k in legs},
        "rule": "the faster partitioning whose gather_check (and self_check, when present) passed; replicated when "
                "neither did"}
    return line


ROTATING_KEYS = ("value", "unit", "ms_per_step", "steps", "path_hbm_frac", "amdahl_bound_speedup", "per_rank",
                 "rotating_owner", "channel_stage_value", "gather_check", "self_check", "pcie_inclusive", "stages", "roofline")


def main():
    args = parse_args()
    if args.gpus < 1:
        die("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                      # does not return
    run = Run(args)
    if run.world != args.gpus:
        # a launcher's world that is not what --gpus asked for is a mis-launch, whichever way round: refuse it rather
        # than print a line whose n_gpus differs from the request
        die("bench.py --gpus %d runs inside a world of %d rank(s): launch `python bench.py --gpus N` (it starts its own "
            "ranks) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, run.world),
            gpus=args.gpus, world=run.world, rank=run.rank)
    torch.cuda.set_device(run.local)
    if run.multi:
        run.start_group()

    from radiocore._internal import hip
    lib = hip.lib()
    hip.torch()

    N, C, B, A, raster, kind = CONFIGS[args.config]
    x, centres, f_in = synth_wideband_on_device(N, C, B, raster, kind, lib, hip)

    legs = ["replicated", "rotating"] if (args.parallelism == "both" and run.world > 1) else \
           ["replicated" if args.parallelism == "both" else args.parallelism]
    n1, refs = None, {}
    if run.self_check:
        # every rank is up (start_group ran a collective); rank 0 now works alone, the others wait for it
        run.phase["leg"] = "one-rank leg"
        if run.rank == 0:
            n1, refs = measure_one_rank(run, x, centres, f_in)
        run.progress("barrier after the one-rank leg")
        run.dist.barrier()
    run.phase["leg"] = legs[0]
    result, (roll_a, bw_a) = measure_partitioning(run, x, centres, f_in, legs[0], primary=True, ref=refs.get(legs[0]))
    rank, world, multi = run.rank, run.world, run.multi
    if multi and rank == 0:
        # against ONE GPU in the same launch (null when the one-rank leg was skipped)
        result["n1"] = n1
        result["speedup_vs_n1"] = round(result["value"] / n1["value"], 4) if n1 and n1["value"] else None
    if len(legs) > 1:
        # the second partitioning in the same launch: its own watchdog budget, its failure is a block of the line
        run.finished = result
        run.phase["leg"] = "rotating (second leg)"
        run.limit = float(os.environ.get("RCFM_BENCH_TIMEOUT_ROTATING", min(run.limit, 120.0)))
        run.progress("second leg: rotating FFT owner")
        try:
            if os.environ.get("RCFM_BENCH_FAIL_ROTATING") == "1" and run.rank == 1:      # tests/test_bench_multirank.py
                raise RuntimeError("injected failure of the rotating leg")
            second, _ = measure_partitioning(run, x, centres, f_in, "rotating", primary=False, ref=refs.get("rotating"))
            if rank == 0:
                result = publish_faster(result, second, n1)
        except Exception as e:
            # the peers are somewhere inside the leg's collectives: no way to tell them -- publish and leave
            run.fail("rotating leg raised %s: %s" % (type(e).__name__, str(e)[:300]))
        run.finished = None
        run.phase["leg"] = "done"

    del refs
    ms_per_step = result["ms_per_step"]
    if rank == 0 and world == 1 and args.config == "cfg4" and not args.no_extras and not multi:
        # the other GPU configurations of BASELINE.json on the same box, outside the headline's timed region
        run.progress("other configurations")
        if "pipelined" not in result:      # (--placement-sets 1: no pair of sets to run lanes on)
            result["pipelined"] = measure_lanes(lib, hip, x, roll_a, bw_a, N, C, B, A, kind, args.steps, args.warmup,
                                                ms_per_step * 1e-3)
        surface4 = measure_surface("cfg4", x, centres, args.steps, args.warmup, ms_per_step * 1e-3)
        del x
        torch.cuda.empty_cache()
        result["other_configs"] = {
            "cfg3": measure_config("cfg3", lib, hip, 50, 5),
            "cfg5": measure_config("cfg5", lib, hip, 20, 3),
            "cfg2_batched": measure_batched_cfg2(lib, hip, 10, 2),
            "geo256k": measure_config("geo256k", lib, hip, 10, 2, with_surface=False),
            "geo200k": measure_config("geo200k", lib, hip, 10, 2, with_surface=False),
            "geo250k": measure_config("geo250k", lib, hip, 10, 2, with_surface=False),
            "path_map": measure_path_map(lib, hip),
            "nbmfm": measure_config("nbmfm", lib, hip, 10, 2, with_surface=False),
            "cfg2_single": measure_cfg2_single(),
            "cfg1_cpu": measure_cfg1_cpu(),
        }
        # the class surface (Tuner.load + Tuner.run_all) against the raw ABI calls of the timed region
        result["surface"] = {"cfg4": surface4, "cfg5": result["other_configs"]["cfg5"].pop("surface"),
                             "cfg3": result["other_configs"]["cfg3"].pop("surface")}
    if args.profile_all and rank == 0:
        tot = sum(v["ms"] for v in result["stages"].values())
        for k, v in result["stages"].items():
            print("%-16s %9.3f ms  %5d launches  %8.1f us each  %6.2f TB/s algorithmic  %4.1f%%" %
                  (k, v["ms"], v["launches"], 1e3 * v["ms"] / max(v["launches"], 1), v["TBps"], 100 * v["ms"] / tot),
                  file=sys.stderr)
    if rank == 0:
        print(json.dumps(result))
        sys.stdout.flush()
    run.progress("done")
    if multi:
        run.dist.destroy_process_group()


if __name__ == "__main__":
    main()

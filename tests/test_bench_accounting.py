"""CPU: the algorithmic-byte accounting bench.py reports against and the
channel grids of its configs (the padded-span rule of tuner.py:163-174)."""

import sys

from conftest import ROOT

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_path_bytes_match_the_survey_totals():
    N, C, B, A, _raster, kind = bench.CONFIGS["cfg4"]
    assert kind == "WBFM"
    total = bench.path_bytes(N, C, B, A, kind)
    assert abs(total - 21.53e9) < 0.01e9              # 3.84 GB + 1024 x 17.28 MB
    assert abs(total / N - 89.7) < 0.1                # bytes per input sample
    N, C, B, A, _raster, kind = bench.CONFIGS["cfg5"]
    assert abs(bench.path_bytes(N, C, B, A, kind) - 4.84e9) < 0.01e9
    N, C, B, A, _raster, kind = bench.CONFIGS["cfg3"]
    assert abs(bench.path_bytes(N, C, B, A, kind) - 590e6) < 1e6


def test_stage_bytes_cover_the_path():
    """The per-stage table adds up to the per-channel totals (WBFM: 16B tuner + 48B + 40A)."""
    N, C, B, A, _raster, kind = bench.CONFIGS["cfg4"]
    assert bench.stage_bytes("tuner_fft_N", N, B, A, kind) == 16 * N
    assert bench.stage_bytes("tuner_ifft_B", N, B, A, kind) == 16 * B
    wbfm = sum(bench.stage_bytes(k, N, B, A, kind) for k in
               ("pilot_stage", "rfft_B", "ifft_B", "fft_B"))
    assert wbfm <= 48 * B + 20 * B                    # the FFT stages of the chain, each read + write once


def test_channel_grids_fit_the_wideband_buffer():
    for name, (N, C, B, _A, raster, _kind) in bench.CONFIGS.items():
        span = (C - 1) * raster + B
        pad = (-span) % (C * B // C)                  # tuner.py:170: span padded to a multiple of the mean bandwidth
        assert span + pad <= N, name


def test_read_only_bytes_match_the_survey_totals():
    """SURVEY.md section 8d, read-only column (the north star says "HBM-read roofline")."""
    N, C, B, A, _raster, kind = bench.CONFIGS["cfg4"]
    assert abs(bench.path_read_bytes(N, C, B, A, kind) - 11.55e9) < 0.01e9
    N, C, B, A, _raster, kind = bench.CONFIGS["cfg5"]
    assert abs(bench.path_read_bytes(N, C, B, A, kind) - 2.70e9) < 0.01e9
    N, C, B, A, _raster, kind = bench.CONFIGS["cfg3"]
    assert abs(bench.path_read_bytes(N, C, B, A, kind) - 350e6) < 1e6


def test_gpus_n_never_prints_a_one_gpu_line(tmp_path):
    """bench.py --gpus 2 on a node that does not show two GPUs (this container shows none), and --gpus 2 inside a
    launcher's world of one rank: both refuse with a JSON error on stderr and a non-zero status -- never a contract
    line that says n_gpus: 1 (round-4 verdict, weak #6)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RCFM_BENCH_DEVICE")}
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                             env=clean, capture_output=True, text=True, timeout=240)
        assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        err = json.loads([ln for ln in out.stderr.splitlines() if ln.startswith("{")][-1])
        assert err["gpus"] == 2 and "GPU" in err["error"]
    env = dict(clean, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    err = json.loads([ln for ln in out.stderr.splitlines() if ln.startswith("{")][-1])
    assert err["world"] == 1 and err["gpus"] == 2

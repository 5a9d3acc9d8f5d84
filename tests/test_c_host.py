"""The C ABI without Python: examples/c_host.c drives librcfm.so through include/rcfm.h alone (tuner, demodulator,
feeder, one-rank RCCL gather) -- the loop of examples/multi_fm_server.py:86-106 in C.

CPU: the example compiles and links against the header and the library (plain C, no torch types at the boundary).
GPU: it runs, and the audio it wrote equals the oracle's for the input it wrote (three buffers: de-emphasis state)."""

import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, TOL, have_gpu, rel_err

SRC = os.path.join(ROOT, "examples", "c_host.c")
LIBDIR = os.path.join(ROOT, "radio-core_amd", "radiocore", "_lib")


def _build(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "librcfm.so")):
        import __graft_entry__
        __graft_entry__.build()
    exe = str(tmp_path / "c_host")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + LIBDIR,
                    "-lrcfm", "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe], check=True)
    return exe


def test_c_host_compiles_against_the_header(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
@pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")
def test_c_host_output_matches_the_oracle(tmp_path):
    import radiocore_oracle as oracle
    exe = _build(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, str(tmp_path)], check=True, capture_output=True, text=True, env=env, timeout=600)
    assert out.stdout.strip().endswith("ok"), out.stdout + out.stderr
    # the rotating owner's hand-over and the two-lane loop (two streams, RCFM_OPT_STATE_FENCE) both reproduced the audio file
    assert "rotating owner, buffer 2" in out.stdout and "two lanes, buffer 2 on stream 0: audio identical" in out.stdout
    N, C, B, A, K = 600000, 3, 60000, 12000, 3
    x = np.fromfile(str(tmp_path / "c_host_input.bin"), np.float32).view(np.complex64).reshape(K, N)
    audio = np.fromfile(str(tmp_path / "c_host_audio.bin"), np.float32).reshape(K, C, A, 2)
    ref = oracle.Tuner()
    for f in (100.00e6, 100.05e6, 99.90e6):
        ref.add_channel(f, B, oracle.WBFM(B, A))
    ref.request_bandwidth(float(N))
    assert ref.input_frequency == 99.975e6                   # the geometry the C program hard-codes
    for k in range(K):
        ref.load(x[k])
        for c in ref.channels():
            want = c.demodulator.run(ref.run_pruned(c.index))[0]
            assert rel_err(audio[k, c.index], want) <= TOL, (k, c.index)

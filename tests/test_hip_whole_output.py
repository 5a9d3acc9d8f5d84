"""GPU: whole-output parity at BASELINE's full sizes ("full-size parity is a thin sample").

(a) EVERY channel of cfg4 / cfg5 / cfg3 `run_all()` -- and of two batched geometries off BASELINE's shape: 1024 x WBFM
    256 000 -> 32 000 (the run-time-L2 decimating tile at full size) and 8192 x narrow-band MFM -- against a second HIP
    evaluation that shares no kernel schedule with it:
    the channels added in reverse order (other pair partners, other tiles, other workgroups), other chunking, and the
    kernel chain with every fused form switched off through the API (`Tuner.set_kernel_options` ->
    rcfm_demod_set_option: no LDS-resident chain, no two-transforms-per-tile kernels, complex hand-over instead of the
    phase link).  The two must agree to float32 rounding (5 % of the tolerance) on all C channels.
(b) the oracle (the reference's algorithm: tuner.py:151-161, fm.py:60-67, wbfm.py:66-105) on 64 channels of cfg4 and
    256 of cfg5, demodulated by a pool of worker processes; the cfg5 sample contains the last pairs every persistent
    workgroup of the LDS chain walks (pairs >= npairs - 256).
(c) where the documented off-centre divergence starts: FM.run on carriers 0 .. 5 kHz off the channel centre against
    the oracle's restatement of the reference's float32 unwrap (fm.py:60-65); the offset at which 1e-4 is crossed is
    recorded in INTEGRATION.md.
"""

import multiprocessing as mp
import os

import numpy as np
import pytest

import workloads
from conftest import TOL, have_gpu, rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]

ROUNDING = 0.05 * TOL


@pytest.fixture(scope="module")
def rc():
    import radiocore
    assert radiocore.HasCuda(), "librcfm.so did not load or sees no device"
    return radiocore


@pytest.fixture(scope="module")
def oracle():
    import radiocore_oracle
    return radiocore_oracle


def _demod_worker(args):
    """Runs in a spawned process (no GPU context): the oracle's demodulator on one channel's IQ, two buffers."""
    kind, B, A, iqs = args
    import radiocore_oracle as oracle
    d = getattr(oracle, kind)(B, A)
    ch = 2 if kind == "WBFM" else 1
    return [np.asarray(d.run(iq)).reshape(A, ch).astype(np.float32) for iq in iqs]


def _oracle_audio(kind, B, A, iq_per_channel):
    """{channel: [audio per buffer]} through a pool of oracle workers."""
    items = sorted(iq_per_channel.items())
    workers = max(1, min(32, (os.cpu_count() or 2) // 2, len(items)))
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        res = pool.map(_demod_worker, [(kind, B, A, iqs) for _, iqs in items], chunksize=1)
    return {c: r for (c, _), r in zip(items, res)}


def _setup(rc, name):
    import bench
    import workloads_device
    from radiocore._internal import hip
    N, C, B, A, raster, kind = bench.CONFIGS[name]
    x, centres, f_in = workloads_device.synth_wideband_on_device(N, C, B, raster, kind, hip.lib(), hip)
    return N, C, B, A, kind, x, centres, f_in


def _tuner(rc, kind, centres, B, A, N):
    t = rc.Tuner(cuda=True)
    for f in centres:
        t.add_channel(f, B, getattr(rc, kind)(B, A))
    t.request_bandwidth(float(N))
    return t


def _worst(a, b):
    """Per-channel max|a - b| / max|b| for [C, A, ch] device tensors -> numpy [C]."""
    import torch
    num = torch.amax(torch.abs(a - b), dim=(1, 2))
    den = torch.clamp(torch.amax(torch.abs(b), dim=(1, 2)), min=1e-30)
    return (num / den).cpu().numpy()


@pytest.mark.parametrize("name,buffers,chunk", [("cfg5", 1, 2048), ("cfg4", 2, 512), ("cfg3", 2, 24), ("geo256k", 1, 256),
                                                ("nbmfm", 2, 4096)])
def test_every_channel_against_an_independent_hip_evaluation(rc, name, buffers, chunk):
    import torch
    N, C, B, A, kind, x, centres, f_in = _setup(rc, name)
    fwd = _tuner(rc, kind, centres, B, A, N)
    rev = _tuner(rc, kind, centres[::-1], B, A, N)           # channel i of fwd is channel C-1-i of rev
    rev.set_kernel_options(lds_chain=False, fused_tiles=False, phase_link=False)
    assert fwd.input_frequency == rev.input_frequency == f_in
    for buf in range(buffers):
        xb = x if buf == 0 else torch.roll(x, 1237 * buf)
        fwd.load(xb)
        a = fwd.run_all(numpy_output=False)
        rev.load(xb)
        b = torch.flip(rev.run_all(numpy_output=False, chunk=chunk), dims=(0,))
        assert a.shape == b.shape == (C, A, 2 if kind == "WBFM" else 1)
        assert bool(torch.all(torch.isfinite(a)))
        err = _worst(a, b)
        bad = np.nonzero(err > ROUNDING)[0]
        print(name, "buffer", buf, "worst channel", int(np.argmax(err)), "%.2e" % err.max(), "median %.2e" % np.median(err))
        assert bad.size == 0, (name, buf, bad[:10], err[bad[:10]])
        del a, b


def _sampled_vs_oracle(rc, oracle, name, sample, buffers):
    import torch
    N, C, B, A, kind, x, centres, f_in = _setup(rc, name)
    tuner = _tuner(rc, kind, centres, B, A, N)
    ref = oracle.Tuner()
    for f in centres:
        ref.add_channel(f, B, None)
    ref.request_bandwidth(float(N))
    iqs = {i: [] for i in sample}
    got = []
    x_host = x.cpu().numpy()
    for buf in range(buffers):
        if buf:
            x = torch.roll(x, 1237 * buf)
            x_host = np.roll(x_host, 1237 * buf)
        tuner.load(x)
        audio = tuner.run_all(numpy_output=False)
        got.append(audio[torch.as_tensor(sample, device=audio.device)].cpu().numpy())
        ref.load(x_host)
        for i in sample:
            iqs[i].append(np.asarray(ref.run_pruned(i)))
    want = _oracle_audio(kind, B, A, iqs)
    errs = {}
    for j, i in enumerate(sample):
        errs[i] = max(rel_err(got[buf][j], want[i][buf]) for buf in range(buffers))
    return errs


def _rotating_sample(name, C, anchors, count):
    """`count` channels of C for the oracle: the anchors (ends, both sides of the middle) plus a draw from a generator
    seeded with the digest of the kernel sources (provenance.kernel_source_sha) -- every change of the device code walks
    the oracle over OTHER channels, so successive rounds cover the band instead of re-checking the same 6 %.
    RCFM_ORACLE_SAMPLE_SEED (hex) reproduces a draw.  The seed and the indices go to gpurun_out/oracle_samples.log."""
    import provenance
    from conftest import ROOT
    seed = os.environ.get("RCFM_ORACLE_SAMPLE_SEED") or provenance.kernel_source_sha()
    rng = np.random.default_rng(int(seed, 16))
    rest = [c for c in range(C) if c not in anchors]
    drawn = rng.choice(rest, size=count - len(anchors), replace=False)
    sample = sorted(set(anchors) | {int(c) for c in drawn})
    line = "%s oracle sample: seed %s, %d channels: %s" % (name, seed, len(sample), " ".join(map(str, sample)))
    print(line)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "oracle_samples.log"), "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass
    return sample


def test_cfg4_sixty_four_channels_against_the_oracle(rc, oracle):
    """64 of the 1024 WBFM channels -- the ends, both sides of the middle, and 56 drawn by a seed that follows the kernel
    sources (_rotating_sample) -- two buffers (de-emphasis state)."""
    sample = _rotating_sample("cfg4", 1024, {0, 1, 510, 511, 512, 513, 1022, 1023}, 64)
    assert len(sample) == 64
    errs = _sampled_vs_oracle(rc, oracle, "cfg4", sample, buffers=2)
    worst = max(errs, key=errs.get)
    line = "cfg4: 64 channels against the oracle, worst channel %d: %.2e" % (worst, errs[worst])
    print(line)
    try:
        from conftest import ROOT
        with open(os.path.join(ROOT, "gpurun_out", "oracle_samples.log"), "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass
    assert errs[worst] <= TOL, {k: v for k, v in errs.items() if v > TOL}


def test_cfg5_two_hundred_fifty_six_channels_against_the_oracle(rc, oracle):
    """256 of the 8192 narrow FM channels: the LDS chain is a persistent kernel, workgroup w of G walks pairs
    w, w + G, ...; the last pair of every workgroup's walk is among pairs >= 4096 - G (G <= 256 CUs), i.e. channels
    >= 7680: 128 of those (both members of 64 pairs), 128 spread over the rest."""
    tail = [c for p in range(4096 - 256, 4096, 4) for c in (2 * p, 2 * p + 1)]
    rest = list(range(0, 7680, 60))
    sample = sorted(set(tail) | set(rest))[:256]
    assert len(sample) == 256 and sum(c >= 7680 for c in sample) >= 128
    errs = _sampled_vs_oracle(rc, oracle, "cfg5", sample, buffers=1)
    worst = max(errs, key=errs.get)
    print("cfg5: 256 channels, worst", worst, "%.2e" % errs[worst])
    assert errs[worst] <= TOL, {k: v for k, v in errs.items() if v > TOL}


def test_offset_at_which_the_float32_unwrap_divergence_crosses_the_tolerance(rc, oracle):
    """fm.py:60-65 unwraps the phase in float32; a carrier `offset` Hz off the channel centre accumulates 2 pi offset
    rad over the one-second buffer and `diff` turns the float32 rounding of that phase into noise.  The HIP
    discriminator takes wrapped phase steps and has no accumulated phase.  This locates the
    offset at which the two part by more than the 1e-4 tolerance, for the broadcast geometry 240 000 -> 48 000 and a
    75 kHz deviation.  Below 500 Hz parity must hold; the measured crossing is printed and kept in INTEGRATION.md."""
    B, A = 240000, 48000
    offsets = [0, 100, 200, 300, 500, 750, 1000, 1500, 2000, 3000, 5000]
    rows = []
    for off in offsets:
        iq = workloads.single_channel(B, i=3, stereo=False, noise=0.0, offset=off)
        got = np.asarray(rc.FM(B, A).run(iq))
        want = np.asarray(oracle.FM(B, A).run(iq))
        z = iq.astype(np.complex128)
        d = np.zeros(B)
        d[1:] = np.angle(z[1:] * np.conj(z[:-1])) / np.pi
        truth = oracle.Decimate(B, A).run(d).reshape(A, 1)
        rows.append((off, rel_err(got, want), rel_err(got, truth), rel_err(want, truth)))
    print("offset Hz | HIP vs reference algorithm | HIP vs float64 truth | reference algorithm vs truth")
    for r in rows:
        print("%8d | %.2e | %.2e | %.2e" % r)
    crossing = next((off for off, e, _, _ in rows if e > TOL), None)
    print("first offset above 1e-4:", crossing)
    for off, e, et, _ in rows:
        assert et <= 0.2 * TOL, (off, et)           # the HIP path stays on the float64 truth at every offset
        if off <= 500:
            assert e <= TOL, (off, e)
    assert crossing is None or crossing > 500

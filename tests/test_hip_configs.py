"""GPU: the batched caller (Tuner.run_all -> rcfm_pipeline_run) on every BASELINE.json GPU configuration.

Reference caller: examples/multi_fm_server.py:98-106 (load, then run -> demodulator.run per channel) with
radiocore/analog/fm.py:60-67, mfm.py:62-66, wbfm.py:66-105 behind it.  Reduced sizes run every channel and two
buffers (state carry) against the oracle loop; the full geometries of cfg3 / cfg4 / cfg5 (BASELINE.json
configs[2..4]) check a sample of channels against the oracle fed with the same wideband buffer, plus
size-independent properties (Parseval on the stored spectrum, idempotence of a second identical buffer for the
state-free FM path, bounded clipped output).

Tolerance: max|delta| <= 1e-4 * max|expected| (BASELINE.json north star), float32 end to end.
"""

import ctypes

import numpy as np
import pytest

import workloads
from conftest import TOL, have_gpu, rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]


@pytest.fixture(scope="module")
def rc():
    import radiocore
    assert radiocore.HasCuda(), "librcfm.so did not load or sees no device"
    return radiocore


@pytest.fixture(scope="module")
def oracle():
    import radiocore_oracle
    return radiocore_oracle


def _pair(rc, oracle, kind, centres, B, A, N):
    tuner, ref = rc.Tuner(), oracle.Tuner()
    for f in centres:
        tuner.add_channel(f, B, getattr(rc, kind)(B, A))
        ref.add_channel(f, B, getattr(oracle, kind)(B, A))
    tuner.request_bandwidth(float(N))
    ref.request_bandwidth(float(N))
    assert tuner.input_frequency == ref.input_frequency
    return tuner, ref


# ---- reduced size: every channel, two buffers, ragged chunks --------------------------------------

@pytest.mark.parametrize("kind,chunk", [("FM", 3), ("MFM", 3), ("WBFM", 3), ("FM", 4), ("MFM", 0), ("MFM", 1)])
def test_run_all_every_demodulator(rc, oracle, kind, chunk):
    """Odd channel count; chunk 3 leaves chunks of 3 + 3 + 1 (a lone member in every chunk's last pair and a
    one-channel remainder), chunk 4 gives 4 + 3, chunk 1 runs every pair kernel with a single member.
    The second buffer checks the de-emphasis state carried per channel (first * ch * 50 offsets)."""
    N, B, A, C = 1_200_000, 60000, 12000, 7
    centres = workloads.channel_grid(C, 50000)
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    stereo = kind == "WBFM"
    for buf in range(2):
        x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.35, stereo=stereo)
        x = np.roll(x, 4321 * buf)
        tuner.load(x)
        ref.load(x)
        audio = tuner.run_all(chunk=chunk)
        ch = 2 if stereo else 1
        assert audio.shape == (C, A, ch) and audio.dtype == np.float32
        for c in ref.channels():
            want = np.asarray(c.demodulator.run(ref.run_pruned(c.index))).reshape(A, ch)
            assert rel_err(audio[c.index], want) <= TOL, (kind, buf, c.index)


@pytest.mark.parametrize("A", [12150, 12006])
@pytest.mark.parametrize("kind", ["FM", "MFM"])
def test_run_all_a_not_multiple_of_four(rc, oracle, kind, A):
    """A % 4 != 0 takes MFM off the packed-pair decimation (api.hip run_chunk) onto the generic de-emphasis
    kernels.  A = 12150 = 2 * 3^5 * 5^2 stays on the FFT engine but is no tile-aligned decimation of B = 60000
    (pair FFT -> full spectrum -> unpack -> real-output inverse FFT); A = 12006 has the prime factors 23 and 29:
    every transform of the demodulator goes through rocFFT."""
    N, B, C = 1_200_000, 60000, 3
    centres = workloads.channel_grid(C, 70000)
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    for buf in range(2):
        x = np.roll(workloads.wideband(N, ref.input_frequency, centres, B, gain=0.35, stereo=False), 99 * buf)
        tuner.load(x)
        ref.load(x)
        audio = tuner.run_all()
        for c in ref.channels():
            want = np.asarray(c.demodulator.run(ref.run_pruned(c.index))).reshape(A, 1)
            assert rel_err(audio[c.index], want) <= TOL, (kind, buf, c.index)


@pytest.mark.parametrize("kind,B,A,chunk", [
    ("FM", 12500, 8000, 0), ("MFM", 12500, 8000, 0), ("FM", 12500, 8000, 7), ("MFM", 12500, 6250, 0),
    ("MFM", 12500, 5000, 7), ("FM", 10000, 8000, 7), ("MFM", 10000, 5000, 0), ("MFM", 12000, 8000, 0),
    ("FM", 12000, 6000, 7), ("MFM", 8000, 4000, 7), ("FM", 12500, 4000, 0)])
def test_run_all_narrowband_fm_geometry(rc, oracle, kind, B, A, chunk):
    """cfg5's channel geometry (12.5 kHz channels, 12 kHz raster) and its neighbours at a reduced band: N = 1e6, 81
    channels.  Every (B, A) of lds_chain.hip's table (12 500 -> 8000 / 6250 / 5000, 10 000 -> 8000 / 5000,
    12 000 -> 8000 / 6000, 8000 -> 4000) runs the whole chain of a channel pair in one workgroup, every transform in
    LDS: odd channel count and chunks of 7 leave lone pair members; MFM adds the de-emphasis kernel and its carried state
    (two buffers).  12 500 -> 4000 has no LDS instantiation and stays on the multi-pass launches (B = 100 x 125): both
    must agree with the reference loop (tuner.py:151-161, fm.py:60-67, mfm.py:62-66).  The stage profile says which
    path ran."""
    from radiocore._internal import hip
    lib = hip.lib()
    N, C = 1_000_000, 81
    centres = workloads.channel_grid(C, int(0.96 * B))
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.1, stereo=False, deviation=0.2 * B)
    hip.check(lib.rcfm_profile_enable((1 << lib.rcfm_profile_stage_count()) - 1))
    hip.check(lib.rcfm_profile_reset())
    try:
        for buf in range(2 if kind == "MFM" else 1):
            xb = np.roll(x, 4242 * buf)
            tuner.load(xb)
            ref.load(xb)
            audio = tuner.run_all(chunk=chunk)
            assert audio.shape == (C, A, 1)
            for c in ref.channels():
                iq = ref.run_pruned(c.index)
                if c.index in (0, 40, 80) and buf == 0:
                    assert rel_err(tuner.run(c.index), iq) <= TOL, c.index
                want = np.asarray(c.demodulator.run(iq)).reshape(A, 1)
                assert rel_err(audio[c.index], want) <= TOL, (kind, B, A, buf, c.index, rel_err(audio[c.index], want))
        ran = _stages_run(lib, hip)
    finally:
        hip.check(lib.rcfm_profile_enable(0))
    assert (ran["lds_chain"] > 0) == ((B, A) != (12500, 4000)), ran


def test_lds_chain_pairing_does_not_matter(rc, oracle):
    """The same narrow band through the LDS-resident kernel and, through a sharded call that starts at an odd channel
    index, a different pairing: a channel's audio must
    not depend on which neighbour shares its complex transform beyond float32 rounding."""
    N, B, A, C = 1_000_000, 12500, 8000, 20
    centres = workloads.channel_grid(C, 12000)
    tuner, ref = _pair(rc, oracle, "FM", centres, B, A, N)
    x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.1, stereo=False, deviation=2500.0)
    tuner.load(x)
    full = tuner.run_all()
    tuner.shard(3, 11)                   # channels 3..13: pairs (3,4) (5,6) ... instead of (2,3) (4,5) ...
    tuner.load(x)
    part = tuner.run_all()
    assert part.shape == (11, A, 1)
    for i in range(11):
        assert rel_err(part[i], full[3 + i]) <= 0.05 * TOL, i


@pytest.mark.parametrize("kind,B", [("FM", 375000), ("MFM", 337500)])
def test_run_all_on_a_three_pass_band(rc, oracle, kind, B):
    """Channels wider than 262 144 samples get a THREE-pass inverse FFT in the tuner (375 000 = 40 . 75 . 125,
    337 500 = 60 . 75 . 75: first-pass lengths that are not multiples of 16).  The padded phase rows of the
    narrow-channel path (rcfm_pipeline_run, cfg5) only exist for two-pass plans; these geometries must stay on
    contiguous phases and agree with the reference loop (multi_fm_server.py:100-106, fm.py:60-67)."""
    N, A, C = 1_500_000, 37500, 3
    centres = workloads.channel_grid(C, 380000)
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.4, stereo=False, deviation=0.1 * B)
    for buf in range(2):
        xb = np.roll(x, 313 * buf)
        tuner.load(xb)
        ref.load(xb)
        audio = tuner.run_all()
        assert audio.shape == (C, A, 1)
        for c in ref.channels():
            iq = ref.run_pruned(c.index)
            want = np.asarray(c.demodulator.run(iq)).reshape(A, 1)
            assert rel_err(audio[c.index], want) <= TOL, (kind, buf, c.index, rel_err(audio[c.index], want))


# ---- full geometries --------------------------------------------------------------------------------

def _full_config(rc, oracle, name, sample, buffers=1):
    """Runs BASELINE config `name` at full size on the GPU; returns per sampled channel the worst error of
    the audio against the oracle (over `buffers` consecutive buffers) and the tuner's IQ error."""
    import torch
    import bench
    import workloads_device
    from radiocore._internal import hip
    N, C, B, A, raster, kind = bench.CONFIGS[name]
    lib = hip.lib()
    x, centres, f_in = workloads_device.synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    tuner = rc.Tuner()
    for f in centres:
        tuner.add_channel(f, B, getattr(rc, kind)(B, A))
    tuner.request_bandwidth(float(N))
    assert tuner.input_frequency == f_in
    ref = oracle.Tuner()
    for f in centres:
        ref.add_channel(f, B, None)
    ref.request_bandwidth(float(N))
    demods = {i: getattr(oracle, kind)(B, A) for i in sample}
    ch = 2 if kind == "WBFM" else 1
    errs = {i: 0.0 for i in sample}
    iq_err = 0.0
    x_host = x.cpu().numpy()
    for buf in range(buffers):
        if buf:
            x = torch.roll(x, 1237 * buf)
            x_host = np.roll(x_host, 1237 * buf)
        tuner.load(x)
        audio = tuner.run_all()
        assert audio.shape == (C, A, ch) and audio.dtype == np.float32
        assert np.all(np.isfinite(audio))
        if kind != "FM":
            assert float(np.max(np.abs(audio))) <= 0.999 + 1e-6          # np.clip of mfm.py:65 / wbfm.py:100
        ref.load(x_host)
        for i in sample:
            iq = ref.run_pruned(i)
            if buf == 0:
                iq_err = max(iq_err, rel_err(tuner.run(i), iq))
            want = np.asarray(demods[i].run(iq)).reshape(A, ch)
            errs[i] = max(errs[i], rel_err(audio[i], want))
    # Parseval on the stored wideband spectrum (size-independent property of Tuner.load)
    X = ctypes.c_void_p()
    hip.check(lib.rcfm_tuner_spectrum(tuner._handle.value, ctypes.byref(X)))
    spec = torch.empty(N, dtype=torch.complex64, device="cuda")
    hip.check(lib.rcfm_memcpy_d2d(hip.ptr(spec), X, N * 8, hip.stream()))
    torch.cuda.synchronize()
    e_t = float(torch.sum(x.real.double() ** 2 + x.imag.double() ** 2))
    e_f = float(torch.sum(spec.real.double() ** 2 + spec.imag.double() ** 2)) / N
    assert abs(e_f - e_t) <= 1e-5 * e_t
    return errs, iq_err, audio, tuner, x


def test_cfg3_full_size_mfm(rc, oracle):
    """BASELINE configs[2]: N = 10 000 000 -> 64 x MFM (240 kHz -> 48 kHz), every 7th channel + the ends,
    two buffers (de-emphasis state of the batched FM/MFM pipeline)."""
    sample = sorted(set(range(0, 64, 7)) | {31, 32, 63})
    errs, iq_err, _, _, _ = _full_config(rc, oracle, "cfg3", sample, buffers=2)
    print("cfg3", errs, iq_err)
    assert iq_err <= TOL
    assert max(errs.values()) <= TOL, errs


def test_cfg5_full_size_fm(rc, oracle):
    """BASELINE configs[4]: N = 100 000 000 -> 8192 x FM (12.5 kHz -> 8 kHz) in one launch per stage (the default
    for channels this narrow); channels either side of every 2048-channel boundary and the ends.  A second identical
    load must give bit-identical audio (FM carries no state), in 2048-channel chunks as well."""
    sample = [0, 1, 2047, 2048, 4095, 4096, 6143, 6144, 8190, 8191]
    errs, iq_err, audio, tuner, x = _full_config(rc, oracle, "cfg5", sample)
    print("cfg5", errs, iq_err)
    assert iq_err <= TOL
    assert max(errs.values()) <= TOL, errs
    tuner.load(x)
    assert np.array_equal(tuner.run_all(), audio)
    assert np.array_equal(tuner.run_all(chunk=2048), audio)


def test_cfg4_full_size_wbfm(rc, oracle):
    """BASELINE configs[3] (the bench workload): N = 240 000 000 -> 1024 x WBFM; the ends and the middle (the
    512-channel chunk boundary of explicit chunking), two buffers: the second one exercises the per-channel
    de-emphasis state of the batched pipeline at full size."""
    sample = [0, 511, 512, 1023]
    errs, iq_err, _, _, _ = _full_config(rc, oracle, "cfg4", sample, buffers=2)
    print("cfg4", errs, iq_err)
    assert iq_err <= TOL
    assert max(errs.values()) <= TOL, errs


@pytest.mark.parametrize("kind", ["WBFM", "MFM"])
def test_run_all_on_an_untidy_band(rc, oracle, kind):
    """A wideband buffer that is not the tidy recipe of the other tests: stations of unequal strength (14 dB apart),
    carriers a few tens of Hz off their channel centres (small enough that the reference's float32 unwrap, fm.py:62,
    stays below 2e-5 -- DESIGN.md section 6 covers the large-offset case), a strong station next to a weak one on
    overlapping channels, one channel that holds nothing but noise, more noise overall."""
    N, B, A, C = 2_400_000, 60000, 12000, 9
    centres = workloads.channel_grid(C, 50000)               # 60 kHz channels on a 50 kHz raster: neighbours overlap
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    f_in = ref.input_frequency
    gains = [0.9, 0.2, 0.6, 0.25, 0.0, 1.0, 0.18, 0.7, 0.45]
    offsets = [0, 37, -52, 11, 0, -23, 60, -8, 41]
    rng = np.random.default_rng(99)
    Xw = np.zeros(N, np.complex128)
    kk = np.fft.fftfreq(B, 1.0 / B).astype(np.int64)
    for i, fc in enumerate(centres):
        if gains[i] == 0.0:
            continue                                          # channel 4: noise only
        s = workloads.station_iq(20 + i, B, stereo=(kind == "WBFM"), offset=offsets[i])
        np.add.at(Xw, (kk + int(fc - f_in)) % N, np.fft.fft(s) * (0.3 * gains[i] * N / B))
    x = np.fft.ifft(Xw) + 0.01 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
    x = x.astype(np.complex64)
    ch = 2 if kind == "WBFM" else 1
    for buf in range(2):
        xb = np.roll(x, 7777 * buf)
        tuner.load(xb)
        ref.load(xb)
        audio = tuner.run_all()
        for c in ref.channels():
            iq = ref.run_pruned(c.index)
            assert rel_err(tuner.run(c.index), iq) <= TOL, (kind, buf, c.index)
            want = np.asarray(c.demodulator.run(iq)).reshape(A, ch)
            if gains[c.index] == 0.0:
                # noise only: the phase steps are uniform in (-pi, pi), so the wrap at +-pi (and, for WBFM, the
                # carrier regeneration from a noise "pilot") turns float32 rounding into O(1) differences in the
                # reference itself -- only the linear part (the tuner's IQ, above) is comparable
                assert np.all(np.isfinite(audio[c.index]))
                continue
            assert rel_err(audio[c.index], want) <= TOL, (kind, buf, c.index, rel_err(audio[c.index], want))


@pytest.mark.parametrize("kind", ["MFM", "WBFM"])
def test_off_raster_stations_through_run_all(rc, oracle, kind, monkeypatch):
    """Stations a few hundred Hz off their channel centres (real broadcast bands are never on the tuner's raster).
    The reference unwraps the phase in float32 (fm.py:62): at these offsets the accumulated phase reaches
    2 pi * offset ~ 2000-4000 rad and ITS audio drifts 4e-6 .. 3e-5 of peak away from the float64 evaluation of the
    same formulas -- still inside the 1e-4 tolerance, so parity with the reference must hold, AND the HIP path
    (wrapped phase steps, no accumulated phase) must be the one that is closer to the float64 truth on every
    off-raster channel.  Truth = the oracle with its discriminator replaced by arg(x[t] conj x[t-1]) / pi in
    float64 (everything downstream then runs in float64 too).  DESIGN.md section 6 has the large-offset case."""
    N, B, A, C = 2_400_000, 60000, 12000, 5
    centres = workloads.channel_grid(C, 70000)
    offsets = [310, -450, 0, 620, -170]
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    f_in = ref.input_frequency
    Xw = np.zeros(N, np.complex128)
    kk = np.fft.fftfreq(B, 1.0 / B).astype(np.int64)
    for i, fc in enumerate(centres):
        s = workloads.station_iq(40 + i, B, stereo=(kind == "WBFM"), offset=offsets[i])
        np.add.at(Xw, (kk + int(fc - f_in)) % N, np.fft.fft(s) * (0.3 * N / B))
    rng = np.random.default_rng(3)
    x = (np.fft.ifft(Xw) + 0.003 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
    ch = 2 if kind == "WBFM" else 1
    tuner.load(x)
    ref.load(x)
    audio = tuner.run_all()
    iqs = [ref.run_pruned(i) for i in range(C)]
    want_ref = [np.asarray(c.demodulator.run(iqs[c.index])).reshape(A, ch) for c in ref.channels()]

    def truth_discriminator(sig):
        z = np.asarray(sig).astype(np.complex128)
        d = np.zeros(len(z))
        d[1:] = np.angle(z[1:] * np.conj(z[:-1])) / np.pi
        return d

    monkeypatch.setattr(oracle, "discriminator", truth_discriminator)
    truth = [np.asarray(getattr(oracle, kind)(B, A).run(iqs[i])).reshape(A, ch) for i in range(C)]
    for i in range(C):
        vs_ref, vs_truth, ref_vs_truth = rel_err(audio[i], want_ref[i]), rel_err(audio[i], truth[i]), \
            rel_err(want_ref[i], truth[i])
        print(kind, i, offsets[i], "hip-ref %.2e  hip-truth %.2e  ref-truth %.2e" % (vs_ref, vs_truth, ref_vs_truth))
        assert vs_ref <= TOL, (kind, i, vs_ref)
        assert vs_truth <= 0.1 * TOL, (kind, i, vs_truth)
        if offsets[i]:
            assert vs_truth < ref_vs_truth, (kind, i, vs_truth, ref_vs_truth)


@pytest.mark.parametrize("kind", ["MFM", "WBFM"])
def test_per_channel_and_batched_calls_share_one_state_per_channel(rc, oracle, kind):
    """In the reference the de-emphasis state lives in the demodulator object a Channel carries
    (deemphasis.py:48-49,64; wbfm.py:53-57), so it does not matter which caller runs a channel.  Here the batched
    call keeps its states in one [C][ch][50] buffer and every channel's demodulator is bound to its slot
    (rcfm_demod_bind_state): the reference's loop (multi_fm_server.py:100-106) and run_all / run_each / a second
    batched handle (another chunk size) may alternate from buffer to buffer, and every buffer must equal what the
    reference's loop gives for that buffer -- which depends on the state all previous buffers left."""
    N, B, A, C = 1_200_000, 60000, 12000, 5
    centres = workloads.channel_grid(C, 50000)
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    ch = 2 if kind == "WBFM" else 1
    x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.35, stereo=(kind == "WBFM"))
    # one warm-up through the per-channel caller BEFORE the batched handle exists: the history it leaves in the
    # demodulator objects must move into the batched buffer when the binding happens
    callers = ["loop", "all", "loop", "each", "all3", "loop", "all"]
    for buf, how in enumerate(callers):
        xb = np.roll(x, 1711 * buf) * np.float32(1.0 - 0.05 * buf)
        tuner.load(xb)
        ref.load(xb)
        if how == "loop":
            got = [np.asarray(c.demodulator.run(tuner.run(c.index))).reshape(A, ch) for c in tuner.channels()]
        elif how == "each":
            got = [np.asarray(a).reshape(A, ch) for a in tuner.run_each()]
        else:
            got = tuner.run_all(chunk=3 if how == "all3" else 0)
        for c in ref.channels():
            want = np.asarray(c.demodulator.run(ref.run_pruned(c.index))).reshape(A, ch)
            assert rel_err(got[c.index], want) <= TOL, (kind, buf, how, c.index, rel_err(got[c.index], want))
    # the state really is one: what the demodulator object reports is its slot of the batched buffer
    one = tuner.channels()[2].demodulator.state()
    tuner.channels()[2].demodulator.reset()
    assert not np.array_equal(tuner.channels()[2].demodulator.state(), one)
    tuner.load(x)
    ref.load(x)
    ref.channels()[2].demodulator = getattr(oracle, kind)(B, A)          # the reference's "fresh demodulator"
    got = tuner.run_all()
    for c in ref.channels():
        want = np.asarray(c.demodulator.run(ref.run_pruned(c.index))).reshape(A, ch)
        assert rel_err(got[c.index], want) <= TOL, (kind, "after reset", c.index)


def test_run_each_on_a_mixed_band(rc, oracle):
    """The reference's loop accepts any mix of channels (examples/multi_fm_server.py:100-106 just calls each
    channel's demodulator): broadcast WBFM next to narrow MFM / FM channels of other bandwidths and audio rates, the
    same geometry appearing in two separate groups, an odd-sized group (a lone member in its last pair).
    Tuner.run_each() must return what that loop returns, for two buffers (de-emphasis state per channel index)."""
    N = 2_400_000
    plan = [("WBFM", -900_000, 120000, 24000), ("WBFM", -700_000, 120000, 24000),
            ("MFM", -500_000, 60000, 12000), ("MFM", -400_000, 60000, 12000),
            ("FM", -300_000, 30000, 6000), ("FM", -250_000, 30000, 6000), ("FM", -200_000, 30000, 6000),
            ("WBFM", 100_000, 120000, 24000), ("MFM", 400_000, 60000, 8000), ("FM", 900_000, 60000, 12000)]
    f0 = 100_000_000
    tuner, ref = rc.Tuner(), oracle.Tuner()
    for kind, off, B, A in plan:
        tuner.add_channel(f0 + off, B, getattr(rc, kind)(B, A))
        ref.add_channel(f0 + off, B, getattr(oracle, kind)(B, A))
    tuner.request_bandwidth(float(N))
    ref.request_bandwidth(float(N))
    assert tuner.input_frequency == ref.input_frequency
    with pytest.raises(ValueError, match="one demodulator class and geometry"):
        tuner.load(np.zeros(N, np.complex64))
        tuner.run_all()
    Xw = np.zeros(N, np.complex128)
    for i, (kind, off, B, A) in enumerate(plan):
        dev = 75e3 * B / 240000.0 if kind == "WBFM" else 0.15 * B
        s = workloads.station_iq(30 + i, B, deviation=dev, stereo=(kind == "WBFM"))
        kk = np.fft.fftfreq(B, 1.0 / B).astype(np.int64)
        np.add.at(Xw, (kk + int(f0 + off - ref.input_frequency)) % N, np.fft.fft(s) * (0.3 * N / B))
    rng = np.random.default_rng(5)
    x = (np.fft.ifft(Xw) + 0.003 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
    for buf in range(2):
        xb = np.roll(x, 2468 * buf)
        tuner.load(xb)
        ref.load(xb)
        got = tuner.run_each()
        assert len(got) == len(plan)
        for c in ref.channels():
            want = np.asarray(c.demodulator.run(ref.run_pruned(c.index)))
            assert got[c.index].shape == want.shape, (c.index, got[c.index].shape, want.shape)
            assert rel_err(got[c.index], want) <= TOL, (buf, c.index, plan[c.index])


# ---- geometries other than BASELINE's: the fused chain is not benchmark-shaped ---------------------------------

def _stages_run(lib, hip):
    """{stage name: launches} from librcfm's stage profile (rcfm_profile_*)."""
    out = {}
    for st in range(lib.rcfm_profile_stage_count()):
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        hip.check(lib.rcfm_profile_read(st, ctypes.byref(ms), ctypes.byref(cnt)))
        out[lib.rcfm_profile_stage_name(st).decode()] = int(cnt.value)
    return out


@pytest.mark.parametrize("kind,B,A", [
    ("WBFM", 256000, 32000),   # the reference's own benchmark shape (tests/benchmark.py:85): 500 x 512, L2 = 64
    ("WBFM", 250000, 50000),   # 500 x 500, L2 = 100
    ("WBFM", 200000, 40000),   # 400 x 500, L2 = 100
    ("WBFM", 200000, 50000),   # 500 x 400, L2 = 100 (the planner's order 400 x 500 would need L2 = 125: swapped)
    ("WBFM", 240000, 24000),   # 480 x 500, L2 = 50
    ("WBFM", 192000, 48000),   # 480 x 400, L2 = 100
    ("WBFM", 250000, 48000),   # the reference's own single-station example (examples/receive_fm.py:18-19): 500 x 500, L2 = 96
    ("MFM", 250000, 48000),
    ("FM", 250000, 48000),
    ("MFM", 256000, 32000),
    ("FM", 200000, 40000),
    ("MFM", 25000, 8000),      # 25 kHz narrow-band channels: 100 x 250, L2 = 80
    ("FM", 20000, 8000),       # 100 x 200, L2 = 80
])
def test_fused_chain_on_other_geometries(rc, oracle, kind, B, A):
    """wbfm.py:32-59 / mfm.py:24-40 accept any sizes.  For every two-pass B whose first-pass length n_1 divides A
    into an even L2 = A / n_1 the audio decimation runs between FFT_B's last pass and IFFT_A's first on one tile
    (k_fft_tile2_decim for the instantiated pairs, k_fft_tile2_decim_rt for the others) and, for WBFM, the pilot
    chain runs as two-transform tiles: the stage profile must show the fused launches (no separate `ifft_A` /
    `audio_spectrum` stage for WBFM, no `audio_spectrum` for FM / MFM) and the audio must match the reference loop
    (multi_fm_server.py:100-106), two buffers, odd channel count."""
    from radiocore._internal import hip
    lib = hip.lib()
    C = 3
    raster = int(B * 1.25)
    N = 10 * raster
    centres = workloads.channel_grid(C, raster)
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    stereo = kind == "WBFM"
    ch = 2 if stereo else 1
    dev = None if B >= 100000 else 0.2 * B
    hip.check(lib.rcfm_profile_enable((1 << lib.rcfm_profile_stage_count()) - 1))
    hip.check(lib.rcfm_profile_reset())
    try:
        for buf in range(2):
            x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.4, stereo=stereo, deviation=dev)
            x = np.roll(x, 911 * buf)
            tuner.load(x)
            ref.load(x)
            audio = tuner.run_all()
            assert audio.shape == (C, A, ch)
            for c in ref.channels():
                want = np.asarray(c.demodulator.run(ref.run_pruned(c.index))).reshape(A, ch)
                assert rel_err(audio[c.index], want) <= TOL, (kind, B, A, buf, c.index, rel_err(audio[c.index], want))
        ran = _stages_run(lib, hip)
    finally:
        hip.check(lib.rcfm_profile_enable(0))
    print(kind, B, A, {k: v for k, v in ran.items() if v})
    assert ran["audio_spectrum"] == 0, ran            # the decimation never ran as a kernel of its own
    if kind == "WBFM":
        assert ran["ifft_A"] == 0 and ran["fft_B"] > 0 and ran["hilbert_mask"] == 0 and ran["stereo_mix"] == 0, ran


PATH_MAP_B = (240000, 250000, 256000)
PATH_MAP_A = (32000, 44100, 48000)


@pytest.mark.parametrize("A", PATH_MAP_A)
@pytest.mark.parametrize("B", PATH_MAP_B)
def test_path_map_cells_against_the_oracle(rc, oracle, B, A):
    """Every cell of bench.py's `other_configs.path_map` (WBFM, B in {240 000, 250 000, 256 000} x A in {32 000, 44 100,
    48 000}) at reduced N against the reference loop, two buffers: whichever route a geometry takes -- two-transform
    tiles, the run-time decimating tile, separate transforms, or rocFFT for every demodulator transform when A has a
    prime factor above 5 (44 100 = 2^2 3^2 5^2 7^2) -- the audio is the reference's."""
    from radiocore._internal import hip
    lib = hip.lib()
    C = 3
    raster = int(B * 1.25)
    N = 10 * raster
    centres = workloads.channel_grid(C, raster)
    tuner, ref = _pair(rc, oracle, "WBFM", centres, B, A, N)
    hip.check(lib.rcfm_profile_enable((1 << lib.rcfm_profile_stage_count()) - 1))
    hip.check(lib.rcfm_profile_reset())
    try:
        for buf in range(2):
            x = np.roll(workloads.wideband(N, ref.input_frequency, centres, B, gain=0.4, stereo=True), 733 * buf)
            tuner.load(x)
            ref.load(x)
            audio = tuner.run_all()
            for c in ref.channels():
                want = np.asarray(c.demodulator.run(ref.run_pruned(c.index))).reshape(A, 2)
                assert rel_err(audio[c.index], want) <= TOL, (B, A, buf, c.index, rel_err(audio[c.index], want))
        ran = _stages_run(lib, hip)
    finally:
        hip.check(lib.rcfm_profile_enable(0))
    print("path_map cell", B, A, {k: v for k, v in ran.items() if v})
    # never the rocFFT route (hilbert_mask / stereo_mix are stages of that route only): 44 100 = 2^2 3^2 5^2 7^2 runs the
    # fused pilot chain and its A-point transforms through the engine's radix-7 butterfly since round 6
    assert ran["hilbert_mask"] == 0 and ran["stereo_mix"] == 0, ran
    if A % 7 == 0:
        assert ran["ifft_A"] > 0, ran                 # 480 does not divide 44 100: no decimating tile, a separate IFFT_A


@pytest.mark.parametrize("narrow", [0, 2])
@pytest.mark.parametrize("kind,B,A", [("FM", 60000, 12000), ("MFM", 60000, 12000), ("WBFM", 60000, 12000),
                                      ("WBFM", 240000, 48000), ("WBFM", 256000, 32000), ("MFM", 375000, 37500)])
def test_run_all_with_both_tile_widths(rc, oracle, kind, B, A, narrow):
    """The tile kernels exist twice (csrc/tile_ns.h): 16 lines per tile for batches, 8 lines for a handful of channels
    (chosen per launch: fewer than two 16-line tiles per CU).  Reduced-size tests would only ever see the narrow ones,
    so this runs both on purpose (Tuner.set_kernel_options(narrow_tiles=0 / 2) -> rcfm_demod_set_option): every
    demodulator, the cfg4 geometry, the reference's benchmark geometry (256 000 -> 32 000: the run-time-L2 decimating
    kernel) and a three-pass band -- against the reference loop (multi_fm_server.py:100-106), odd channel count, two
    buffers."""
    C = 5 if B <= 60000 else 3
    raster = int(1.2 * B)
    N = (C + 1) * raster
    centres = workloads.channel_grid(C, raster)
    tuner, ref = _pair(rc, oracle, kind, centres, B, A, N)
    tuner.set_kernel_options(narrow_tiles=narrow)
    stereo = kind == "WBFM"
    ch = 2 if stereo else 1
    for buf in range(2):
        x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.4, stereo=stereo,
                               deviation=None if B < 300000 else 0.1 * B)
        x = np.roll(x, 517 * buf)
        tuner.load(x)
        ref.load(x)
        audio = tuner.run_all(chunk=2 if narrow else 0)
        for c in ref.channels():
            want = np.asarray(c.demodulator.run(ref.run_pruned(c.index))).reshape(A, ch)
            assert rel_err(audio[c.index], want) <= TOL, (kind, B, A, narrow, buf, c.index, rel_err(audio[c.index], want))


def test_cfg4_shards_reproduce_the_whole_band_bit_for_bit_under_both_wideband_plans():
    """What bench.py's N > 1 `self_check` relies on, on ONE GPU at cfg4's full size: a rank that declares an eighth of
    the channels (rcfm_tuner_shard: a row window of the wideband FFT's last pass) and runs only those computes exactly
    the audio a one-GPU run of all 1024 channels computes for them -- with the handle's default wideband plan (the aligned
    order in the padded-rows layout: what the replicated partitioning runs) and with RCFM_TUNER_OPT_ALIGNED_PLAN = 0 (the
    default order: what a rotating owner runs into an attached slot).  The two plans differ from each other in rounding
    only (<= 2e-6 of peak), which is why bench.py keeps one reference per plan."""
    import torch
    import bench
    import workloads_device
    from radiocore._internal import hip
    lib = hip.lib()
    N, C, B, A, raster, kind = bench.CONFIGS["cfg4"]
    x, centres, f_in = workloads_device.synth_wideband_on_device(N, C, B, raster, kind, lib, hip)
    rolls = (ctypes.c_int64 * C)(*[int(f_in - f) for f in centres])
    bws = (ctypes.c_int32 * C)(*([B] * C))
    s = hip.stream()

    def run(aligned, lo, cnt):
        t, d = ctypes.c_void_p(), ctypes.c_void_p()
        hip.check(lib.rcfm_tuner_create(N, C, rolls, bws, ctypes.byref(t)))
        hip.check(lib.rcfm_tuner_set_option(t, hip.RCFM_TUNER_OPT_ALIGNED_PLAN, aligned))
        hip.check(lib.rcfm_tuner_shard(t, lo, cnt))
        hip.check(lib.rcfm_demod_create(2, C, B, A, ctypes.c_double(75e-6), 0, ctypes.byref(d)))
        audio = torch.empty((cnt, A, 2), dtype=torch.float32, device="cuda")
        hip.check(lib.rcfm_tuner_load(t, hip.ptr(x), s))
        hip.check(lib.rcfm_pipeline_run(t, d, lo, cnt, hip.ptr(audio), s))
        torch.cuda.synchronize()
        hip.check(lib.rcfm_demod_destroy(d))
        hip.check(lib.rcfm_tuner_destroy(t))
        return audio

    whole = {}
    for aligned in (1, 0):
        whole[aligned] = run(aligned, 0, C)
        assert float(whole[aligned].abs().amax()) > 1e-3
        for lo in (0, 384, 896):                    # first, a middle and the last eighth (the band's ends wrap)
            part = run(aligned, lo, 128)
            assert torch.equal(part, whole[aligned][lo:lo + 128]), (aligned, lo)
            del part
    peak = float(whole[1].abs().amax())
    diff = float((whole[1] - whole[0]).abs().amax())
    assert diff <= 2e-5 * peak, (diff, peak)      # same transform, another order of the passes: rounding only

"""GPU: the HIP path (through the C ABI) against the golden vectors and the oracle.

Tolerance: max|delta| <= 1e-4 * max|expected| (BASELINE.json north star), float32
arithmetic end to end.
"""

import ctypes

import numpy as np
import pytest

import golden_cases as gc
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


def _check(cases, tol=TOL):
    bad = [(n, e) for (n, e) in cases if not e <= tol]
    print(cases)
    assert not bad, bad


# ---- golden vectors captured from the reference ------------------------------

def test_golden_decimate(rc, golden):
    _check(gc.decimate_cases(rc, golden("decimate")))


def test_golden_bandpass(rc, golden):
    _check(gc.bandpass_cases(rc, golden("bandpass")))


def test_golden_deemphasis(rc, golden):
    _check(gc.deemphasis_cases(rc, golden("deemphasis")))


def test_golden_pll(rc, golden):
    _check(gc.pll_cases(rc, golden("pll")))


def test_golden_fm(rc, golden):
    _check(gc.fm_cases(rc, golden("fm")))


def test_fm_off_centre_carrier_follows_the_mathematics(rc, golden):
    """Known, documented divergence: the reference unwraps the phase in float32 (fm.py:62), so
    with a carrier 5 / 50 kHz off the channel centre ITS output is 2.8e-4 / 2.2e-3 away from the float64 evaluation
    of fm.py:60-67.  The HIP discriminator takes wrapped phase steps and has no accumulated phase: it must sit
    within 1e-5 of the float64 truth, and its distance from the reference must be the reference's own error."""
    res = gc.fm_offcentre_cases(rc, golden("fm_offcentre"))
    print(res)
    for off, (vs_ref, vs_truth, ref_vs_truth) in res.items():
        assert vs_truth <= 0.1 * TOL, (off, vs_truth)
        assert ref_vs_truth > TOL
        assert abs(vs_ref - ref_vs_truth) <= 0.05 * ref_vs_truth, (off, vs_ref, ref_vs_truth)


def test_golden_mfm(rc, golden):
    _check(gc.mfm_cases(rc, golden("mfm")))


def test_golden_wbfm(rc, golden):
    _check(gc.wbfm_cases(rc, golden("wbfm")))


def test_golden_cd_rate_audio_and_the_reference_example_geometry(rc, golden):
    """240 000 -> 44 100 (every A-point transform through the engine's radix-7 butterfly) and 250 000 -> 48 000
    (examples/receive_fm.py:18-19), WBFM and MFM, against outputs of the reference itself."""
    _check(gc.geo_44100_cases(rc, golden("wbfm_44100")))


def test_golden_wbfm_ill_conditioned(rc, golden):
    # see tests/golden_cases.py: conditioning ~1.3e4 at one sample
    _check(gc.wbfm_illcond_case(rc, golden("wbfm")), tol=1e-3)


def test_golden_tuner_server_loop(rc, golden):
    _check(gc.tuner_cases(rc, golden("tuner")))


def test_golden_tuner_odd(rc, golden):
    _check(gc.tuner_odd_cases(rc, golden("tuner_odd")))


# ---- batched path vs oracle ----------------------------------------------------

@pytest.mark.parametrize("kind", ["FM", "MFM", "WBFM"])
def test_batched_demod_matches_oracle(rc, oracle, kind):
    B, A, C = 60000, 12000, 5
    x = np.stack([workloads.single_channel(B, i=10 + i) for i in range(C)])
    dev = getattr(rc, kind)(B, A, batch=C, chunk=2)        # 2 + 2 + 1: exercises the remainder plan
    refs = [getattr(oracle, kind)(B, A) for _ in range(C)]
    for buf in range(2):                                   # second buffer checks the state carry
        xb = np.roll(x, 777 * buf, axis=1)
        got = dev.run(xb)
        assert got.shape == (C, A, dev.channels)
        for i in range(C):
            want = refs[i].run(xb[i]).reshape(A, dev.channels)
            assert rel_err(got[i], want) <= TOL, (kind, buf, i)


def test_run_all_equals_per_channel_loop(rc, oracle):
    """Tuner.run_all() == the reference's caller loop (multi_fm_server.py:98-106)."""
    N, B, A, C = 1200000, 60000, 12000, 7
    centres = workloads.channel_grid(C, 50000)
    tuner = rc.Tuner()
    ref = oracle.Tuner()
    for f in centres:
        tuner.add_channel(f, B, rc.WBFM(B, A))
        ref.add_channel(f, B, oracle.WBFM(B, A))
    tuner.request_bandwidth(float(N))
    ref.request_bandwidth(float(N))
    assert tuner.input_frequency == ref.input_frequency
    for buf in range(2):
        x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.35)
        x = np.roll(x, 4321 * buf)
        keep = x.copy()
        tuner.load(x)
        ref.load(x)
        assert np.array_equal(x, keep)                     # inputs are never modified
        audio = tuner.run_all(chunk=3)
        assert audio.shape == (C, A, 2) and audio.dtype == np.float32
        for ch in ref.channels():
            iq = ref.run_pruned(ch.index)
            assert rel_err(tuner.run(ch.index), iq) <= TOL
            want = ch.demodulator.run(iq)[0]
            assert rel_err(audio[ch.index], want) <= TOL, (buf, ch.index)


def test_narrow_channels_take_the_haloed_gather(rc, oracle):
    """Channels much narrower than the band (cfg4's shape, scaled down): the tuner's gather reads the
    haloed spectrum without wrap-around and windows by the cosine series.  The centre channel straddles
    bin 0 (negative bins come from the left halo), odd channel count -> a lone last pair."""
    N, B, A, C = 6_000_000, 60000, 12000, 5
    centres = workloads.channel_grid(C, 75000)
    tuner = rc.Tuner()
    ref = oracle.Tuner()
    for f in centres:
        tuner.add_channel(f, B, rc.WBFM(B, A))
        ref.add_channel(f, B, oracle.WBFM(B, A))
    tuner.request_bandwidth(float(N))
    ref.request_bandwidth(float(N))
    x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.35)
    tuner.load(x)
    ref.load(x)
    audio = tuner.run_all()
    for ch in ref.channels():
        iq = ref.run_pruned(ch.index)
        assert rel_err(tuner.run(ch.index), iq) <= TOL, ch.index
        assert rel_err(audio[ch.index], ch.demodulator.run(iq)[0]) <= TOL, ch.index


def test_sharded_tuner_keeps_what_its_channels_read(rc, oracle):
    """Tuner.shard(first, count): the wideband FFT stores only the rows of the spectrum the declared channels
    read; their outputs do not change (every rank of a multi-GPU run works like this)."""
    N, B, A, C = 6_000_000, 60000, 12000, 8
    centres = workloads.channel_grid(C, 700000)          # spread over the band: a shard needs a fraction of it
    full, part = rc.Tuner(), rc.Tuner()
    ref = oracle.Tuner()
    for f in centres:
        full.add_channel(f, B, rc.MFM(B, A))
        part.add_channel(f, B, rc.MFM(B, A))
        ref.add_channel(f, B, None)
    for t in (full, part, ref):
        t.request_bandwidth(float(N))
    x = workloads.wideband(N, ref.input_frequency, centres, B, gain=0.35)
    ref.load(x)
    for first, count in ((0, 3), (5, 3), (3, 2)):
        part.shard(first, count)
        part.load(x)
        full.load(x)
        for i in range(first, first + count):
            iq = ref.run_pruned(i)
            assert rel_err(part.run(i), iq) <= TOL, (first, i)
            assert np.array_equal(part.run(i), full.run(i)), (first, i)
    with pytest.raises(IndexError):
        part.shard(6, 3)
    # the range that counts is the one the spectrum was LOADED for: re-declaring the shard without a new load does
    # not make the other rows appear (rcfm_tuner_run checks against the loaded range)
    part.shard(0, 3)
    part.load(x)
    part.shard(5, 3)
    with pytest.raises(RuntimeError, match="outside the shard"):
        part.run(6)
    assert np.array_equal(part.run(1), full.run(1))
    part.load(x)                                   # now the new range is in force
    assert np.array_equal(part.run(6), full.run(6))
    with pytest.raises(RuntimeError, match="outside the shard"):
        part.run(1)


def test_tuner_spectrum_bins(rc, golden):
    """Tuner.load keeps FFT_N(x): spot bins against the reference's."""
    g = golden("tuner")
    import radiocore_oracle as oracle
    specs = gc.tuner_specs(oracle)
    t = rc.Tuner(cuda=True)
    for (f, bw, _cls) in specs:
        t.add_channel(f, bw, None)
    t.request_bandwidth(float(gc.TUNER_N))
    x = workloads.wideband(gc.TUNER_N, t.input_frequency, [s[0] for s in specs], gc.TUNER_B, gain=0.4)
    x = np.roll(x, 12345)
    g.check_input("in1", x)
    t.load(x)
    from radiocore._internal import hip
    X = ctypes.c_void_p()
    hip.check(hip.lib().rcfm_tuner_spectrum(t._handle.value, ctypes.byref(X)))
    host = np.zeros(gc.TUNER_N, np.complex64)
    hip.check(hip.lib().rcfm_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), X, host.nbytes, hip.stream()))
    hip.check(hip.lib().rcfm_stream_sync(hip.stream()))
    N = gc.TUNER_N
    got = host[[0, 1, 2, 1000, N // 2, N - 1]]
    peak = np.max(np.abs(host))
    assert np.max(np.abs(got - g["spectrum_bins"])) <= TOL * peak


# ---- error behaviour and API details ---------------------------------------------

def test_errors_match_the_reference(rc):
    for cls in (rc.FM, rc.MFM, rc.WBFM):
        with pytest.raises(ValueError, match="input_sig size and input_size mismatch"):
            cls(60000, 12000).run(np.zeros(59999, np.complex64))
    for obj in (rc.Decimate(100, 10), rc.Bandpass(1000, 100, 200), rc.Deemphasis(100)):
        with pytest.raises(ValueError, match="input_sig size and input_size mismatch"):
            obj.run(np.zeros(99, np.float32))
    t = rc.Tuner()
    t.add_channel(1e6, 1000, None)
    with pytest.raises(ValueError, match="is too low"):
        t.request_bandwidth(10.0)
    t.load(np.zeros(1000, np.complex64))
    with pytest.raises(IndexError):
        t.run(3)
    with pytest.raises(ValueError):
        rc.Tuner().reset()


def test_tuner_keeps_its_first_window(rc):
    from test_oracle_golden import _stale_window_scenario
    _stale_window_scenario(rc)


def test_device_output_and_reset(rc):
    import torch
    B, A = 60000, 12000
    x = workloads.single_channel(B, i=3)
    d = rc.WBFM(B, A, cuda=True)
    a0 = d.run(x)
    assert isinstance(a0, np.ndarray)
    dev = d.run(x, numpy_output=False)
    assert isinstance(dev, torch.Tensor) and dev.is_cuda and tuple(dev.shape) == (1, A, 2)
    d.reset()
    assert np.array_equal(d.run(x), a0)                    # same state -> bit-identical output
    st = d.state()
    assert st.shape == (1, 2, 50) and np.all(np.isfinite(st))


def test_zero_input_gives_nan_like_the_reference(rc, oracle):
    """tests/benchmark.py feeds zeros: 0/0 in pll.py:58 makes WBFM output NaN."""
    B, A = 60000, 12000
    z = np.zeros(B, np.complex64)
    with np.errstate(all="ignore"):
        want = oracle.WBFM(B, A).run(z)
    got = rc.WBFM(B, A).run(z)
    assert np.all(np.isnan(want)) and np.all(np.isnan(got))
    assert rel_err(rc.MFM(B, A).run(z), oracle.MFM(B, A).run(z)) <= TOL


# ---- full-size configuration through size-independent properties ------------------

def test_full_size_channel_properties(rc, oracle):
    """BASELINE config 3/4 geometry at reduced channel count: N = 10 MS/s, 240 kHz
    channels -> 48 kHz.  Checks (i) Parseval on the stored spectrum, (ii) every channel
    against the oracle fed with the same spectrum bins, (iii) linearity of the tuner."""
    N, B, A, C = 10_000_000, 240000, 48000, 8
    centres = workloads.channel_grid(C, 200000)
    tuner = rc.Tuner()
    for f in centres:
        tuner.add_channel(f, B, rc.WBFM(B, A))
    tuner.request_bandwidth(float(N))
    x = workloads.wideband(N, tuner.input_frequency, centres, B, gain=0.35)
    tuner.load(x)
    audio = tuner.run_all()
    ref = oracle.Tuner()
    for f in centres:
        ref.add_channel(f, B, None)
    ref.request_bandwidth(float(N))
    ref.load(x)                                             # ~0.3 s of pocketfft
    for i in (0, 3, 7):
        iq = ref.run_pruned(i)
        assert rel_err(tuner.run(i), iq) <= TOL
        want = oracle.WBFM(B, A).run(iq)[0]
        assert rel_err(audio[i], want) <= TOL
    # linearity: tuner(a x) == a tuner(x)
    y1 = tuner.run(2)
    tuner.load(0.5 * x)
    assert rel_err(2.0 * tuner.run(2), y1) <= 1e-5


def test_pinned_buffer_feeds_the_tuner(rc):
    """Buffer(cuda=True) is page-locked host memory; loading the Tuner
    from it gives the same channel as loading from an ordinary array."""
    import torch
    N, B = 600000, 60000
    buf = rc.Buffer(N, dtype=np.complex64, cuda=True)
    assert buf.is_cuda and buf.data.dtype == np.complex64 and len(buf) == N
    assert torch.from_numpy(buf.data.view(np.float32)).is_pinned() or buf._owner.is_pinned()
    x = workloads.wideband(N, 100e6, [100e6, 100.1e6], B, gain=0.5)
    with buf.consume() as arr:
        arr[:] = x
    t = rc.Tuner()
    t.add_channel(100e6, B, None)
    t.add_channel(100.1e6, B, None)
    t.request_bandwidth(float(N))
    t.load(buf.data)
    a = t.run(1)
    t.load(x)
    assert np.array_equal(a, t.run(1))
    ring = rc.RingBuffer(N, dtype=np.complex64, cuda=True)
    ring.put(x)
    out = np.zeros(N, np.complex64)
    assert ring.get(out) is True and np.array_equal(out, x)


def test_feeder_overlaps_copies_and_matches_synchronous_load(rc):
    """SURVEY.md section 8f-1, second half: radiocore.tools.Feeder (rcfm_feeder_*) copies buffer i+1 from
    page-locked memory on its own stream while buffer i is processed; every buffer must come out exactly as
    with the synchronous Tuner.load (reference hand-over: examples/multi_fm_server.py:95-98)."""
    N, B, A, K = 600000, 60000, 12000, 5
    centres = [100e6, 100.1e6, 99.9e6]
    base = workloads.wideband(N, 100e6, centres, B, gain=0.5)
    hosts = []
    for k in range(K):                       # K distinct page-locked buffers
        buf = rc.Buffer(N, dtype=np.complex64, cuda=True)
        buf.data[:] = np.roll(base, 1000 * k) * np.float32(1.0 - 0.1 * k)
        hosts.append(buf)

    def tuner():
        t = rc.Tuner()
        for f in centres:
            t.add_channel(f, B, rc.MFM(B, A))
        t.request_bandwidth(float(N))
        return t

    sync, fed = tuner(), tuner()
    want = []
    for buf in hosts:
        sync.load(buf.data)
        want.append((sync.run(1), sync.run_all()))
    feeder = rc.Feeder(N, dtype=np.complex64, depth=2)
    assert feeder.depth == 2
    feeder.submit(hosts[0].data)
    got = []
    for k in range(K):
        if k + 1 < K:
            feeder.submit(hosts[k + 1].data)          # in flight while buffer k is processed
        with feeder.next() as x:
            fed.load(x)
            got.append((fed.run(1), fed.run_all()))
    for k in range(K):
        assert np.array_equal(got[k][0], want[k][0]), k
        assert np.array_equal(got[k][1], want[k][1]), k
    # pageable memory pinned in place with rcfm_host_register works as a source too
    import ctypes
    from radiocore._internal import hip
    plain = np.ascontiguousarray(hosts[2].data.copy())
    hip.check(hip.lib().rcfm_host_register(ctypes.c_void_p(plain.ctypes.data), plain.nbytes))
    feeder.submit(plain)
    with feeder.next() as x:
        fed.load(x)
        assert np.array_equal(fed.run(1), want[2][0])
    import torch
    torch.cuda.synchronize()
    hip.check(hip.lib().rcfm_host_unregister(ctypes.c_void_p(plain.ctypes.data)))
    # protocol errors are reported, not ignored
    feeder.submit(hosts[0].data)
    feeder.submit(hosts[1].data)
    with pytest.raises(RuntimeError, match="in flight"):
        feeder.submit(hosts[2].data)
    with pytest.raises(ValueError, match="size"):
        feeder.submit(np.zeros(10, np.complex64))


def test_feeder_keeps_temporaries_alive_until_their_copy_has_landed(rc):
    """Feeder.submit() may be handed an array nobody else holds (a roll, a scaled copy): the feeder is then the only
    owner until the DMA has run.  The host queues buffers without ever waiting for the GPU here, so dropping a source
    when its slot is released (host-side bookkeeping) would free memory under a pending copy; sources are dropped
    only once rcfm_feeder_copied() says their copy has completed.  (multi_fm_server.py:95-98 is the hand-over.)"""
    import gc
    import torch
    N, B, A, K = 600000, 60000, 12000, 6
    centres = [100e6, 100.1e6]
    base = workloads.wideband(N, 100e6, centres, B, gain=0.5)
    t = rc.Tuner(cuda=True)
    for f in centres:
        t.add_channel(f, B, rc.FM(B, A))
    t.request_bandwidth(float(N))
    want = []
    for k in range(K):
        t.load(np.roll(base, 777 * k))
        want.append(t.run_all())
    feeder = rc.Feeder(N, dtype=np.complex64, depth=2)
    got = []
    feeder.submit(np.roll(base, 0))                       # temporaries: no reference survives this statement
    for k in range(K):
        if k + 1 < K:
            feeder.submit(np.roll(base, 777 * (k + 1)))
        gc.collect()
        with feeder.next() as x:
            t.load(x)
            got.append(t.run_all(numpy_output=False))      # device result: nothing here waits for the GPU
    torch.cuda.synchronize()
    for k in range(K):
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    feeder._drop_landed()
    assert feeder._sources == [] and feeder._dropped == K   # everything has landed: nothing is kept alive
    feeder.submit(np.roll(base, 5))
    feeder.close()                                          # waits for the copy stream, then frees the slots
    assert feeder._slots == [] and feeder._sources == []
    feeder.close()                                          # idempotent


def test_hilbert_on_two_streams_does_not_share_a_workspace(rc, oracle):
    """rcfm_hilbert caches plans and workspaces per (device, stream, n, C): two PLLs stepping concurrently on
    different streams with the same geometry (pll.py:25-34) must not race on one spectrum buffer."""
    import torch
    n = 240000
    rng = np.random.default_rng(11)
    xs = [rng.standard_normal(n).astype(np.float32) for _ in range(2)]
    want = []
    for x in xs:
        p = oracle.PLL()
        p.step(x)
        want.append(p.image(2))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    plls = [rc.PLL(cuda=True), rc.PLL(cuda=True)]
    dev = [torch.from_numpy(x).cuda() for x in xs]
    torch.cuda.synchronize()
    outs = [None, None]
    for rep in range(20):                                   # interleaved, no synchronisation in between
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                plls[i].step(dev[i])
                outs[i] = plls[i].image(2)
    torch.cuda.synchronize()
    for i in (0, 1):
        assert rel_err(outs[i].cpu().numpy(), want[i]) <= 1e-3   # Im(z^2)/|z^2| of noise: conditioning, not the race
        p = rc.PLL()
        p.step(xs[i])
        assert np.array_equal(outs[i].cpu().numpy(), p.image(2)), i   # bit-equal to the single-stream result


def test_server_loop_example(rc, oracle):
    """examples/multi_fm_pipeline.py: producer thread -> RingBuffer -> Feeder -> Tuner.run_all -> wire frames, three
    seconds; every published message equals the oracle's audio for that second and channel."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("multi_fm_pipeline", os.path.join(ROOT, "examples", "multi_fm_pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seconds, C, rate, B, A = 3, 5, 1_200_000, 60000, 12000
    msgs = mod.run(seconds, C, rate, B, A)
    assert len(msgs) == seconds * C
    centres = workloads.channel_grid(C, 50000)
    ref = oracle.Tuner()
    for f in centres:
        ref.add_channel(f, B, oracle.WBFM(B, A))
    ref.request_bandwidth(float(rate))
    second = workloads.wideband(rate, ref.input_frequency, centres, B, gain=0.3)
    for s in range(seconds):
        ref.load(np.roll(second, 1000 * s))
        for c in ref.channels():
            freq, pcm = msgs[s * C + c.index]
            assert freq == int(c.center_frequency)
            want = c.demodulator.run(ref.run_pruned(c.index))[0]
            assert rel_err(pcm, want) <= TOL, (s, c.index)
    # the same loop with two seconds in flight on alternating streams (radiocore.tools.Lanes): the same messages, bit for bit
    msgs2 = mod.run(seconds, C, rate, B, A, lanes=2)
    assert len(msgs2) == len(msgs)
    for (f1, a1), (f2, a2) in zip(msgs, msgs2):
        assert f1 == f2 and np.array_equal(a1, a2)


@pytest.mark.parametrize("kind", ["WBFM", "MFM"])
def test_single_station_example(rc, oracle, kind):
    """examples/receive_fm_offline.py = the reference's examples/receive_fm.py without SDR and sound card: producer thread
    -> RingBuffer -> complex Decimate(10 000 000 -> 250 000) -> WBFM / MFM(250 000 -> 48 000), the reference's own default
    geometry (receive_fm.py:15-21), two seconds (de-emphasis state): every audio block equals the oracle's."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("receive_fm_offline", os.path.join(ROOT, "examples", "receive_fm_offline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seconds, n, b, a = 2, 10_000_000, 250_000, 48_000
    got = mod.run(seconds, kind, n, b, a)
    assert len(got) == seconds
    decim = oracle.Decimate(n, b)
    demod = getattr(oracle, kind)(b, a)
    for s in range(seconds):
        x = mod.capture(n, b, stereo=(kind == "WBFM"), second=s)
        want = np.asarray(demod.run(decim.run(x)))
        assert got[s].shape == want.shape == ((1, a, 2) if kind == "WBFM" else (a, 1))
        assert rel_err(got[s], want) <= TOL, (kind, s, rel_err(got[s], want))

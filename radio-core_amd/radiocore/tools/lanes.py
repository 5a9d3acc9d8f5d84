"""Lanes: consecutive wideband buffers on alternating HIP streams (no reference counterpart).

The reference's server handles one buffer at a time (examples/multi_fm_server.py:98-106: ``load``, then every channel's
``run`` -> ``demodulator.run``).  On the GPU every stage of that loop is a launch whose last workgroups leave part of the
chip idle, and the wideband FFT of buffer i + 1 does not depend on anything buffer i computes: with two handle sets
(spectrum + workspaces each), each on its own stream, the kernels of the next buffer fill the gaps of the current one.
The ONE thing buffer i + 1 needs from buffer i is the de-emphasis state (deemphasis.py:64; mfm.py:63, wbfm.py:97-100):
the lanes share it, and librcfm orders the launches that touch it across streams (RCFM_OPT_STATE_FENCE), so the audio is
bit-identical to the one-buffer-at-a-time loop.

Measured (`pipelined` blocks of profiles/r04_i_bench.json, same box as the one-lane figure): cfg4 -2 %, cfg5 -2 %, cfg3 (64 channels, launches of a few
tiles per CU) -11 %.  Costs one more spectrum and workspace set per lane (cfg4: 14 GB).
"""

from radiocore._internal import hip
from radiocore.tools.tuner import Tuner

__all__ = ["Lanes"]


class Lanes:
    """``depth`` buffers in flight through one Tuner's channels.

        lanes = Lanes(tuner, depth=2)
        t0 = lanes.submit(buffer0)          # returns at once; load + run_all are queued on lane 0's stream
        t1 = lanes.submit(buffer1)          # lane 1: overlaps buffer0's kernels
        audio0 = lanes.result(t0)           # [C, A, ch] like Tuner.run_all(); waits for that buffer only

    Buffers are demodulated in submission order as far as the per-channel state is concerned; results may be collected
    in any order, each once.  ``tuner`` stays usable on its own between submissions (it is lane 0).
    Submit from ONE host thread (or order the calls yourself): the order of the host calls is the order of the buffers,
    and the fence is armed by whichever call came last.  Before resetting or reading a demodulator's state by hand
    (``demodulator.reset()``), ``drain()`` first.
    """

    def __init__(self, tuner: Tuner, depth: int = 2, timing: bool = False):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self._torch = hip.torch()
        self._timing = bool(timing)     # events that can be timed: overlap_ms() (tests, tuning), not needed otherwise
        self._base = tuner
        tuner._arm_state_fence()
        self._tuners = [tuner] + [tuner._lane_clone() for _ in range(depth - 1)]
        self._streams = [self._torch.cuda.Stream() for _ in range(depth)]
        self._next = 0
        self._pending = {}          # ticket -> (end event, audio tensor, event behind the load = last reader of the input)
        self._spans = {}            # timing=True: ticket -> (start event, end event)

    @property
    def depth(self) -> int:
        return len(self._tuners)

    def submit(self, input_signal, chunk: int = 0, each: bool = False) -> int:
        """Queue Tuner.load + Tuner.run_all of one buffer on the next lane; returns a ticket for ``result``.
        each=True: Tuner.run_each instead (channels of different classes and geometries; ``result`` then returns its
        list of per-channel arrays)."""
        ticket = self._next
        self._next += 1
        k = ticket % len(self._tuners)
        t = self._tuners[k]
        if k:
            t._sync_lane(self._base)            # channels added since the lane was made
        st = self._streams[k]
        st.wait_stream(self._torch.cuda.current_stream())     # whatever produced the buffer
        if isinstance(input_signal, self._torch.Tensor) and input_signal.is_cuda:
            # the caller's tensor is read on the lane's stream: the caching allocator must not hand its memory to
            # current-stream work before the lane's FFT has consumed it
            input_signal.record_stream(st)
        with self._torch.cuda.stream(st):
            start = self._event(st) if self._timing else None
            t.load(input_signal)
            loaded = self._event(st)            # the wideband FFT is the LAST reader of the input buffer
            audio = t._run_each_device() if each else t._run_all_device(chunk)
            ev = self._event(st)
        self._pending[ticket] = (ev, audio, loaded)
        if self._timing:
            self._spans[ticket] = (start, ev)
        return ticket

    def _event(self, stream):
        e = self._torch.cuda.Event(enable_timing=self._timing)
        e.record(stream)
        return e

    def overlap_ms(self, earlier: int, later: int) -> float:
        """timing=True only: how long before the END of buffer `earlier` the lane of buffer `later` started working
        (positive: the two buffers overlapped on the device; negative: `later` started that long after `earlier` had
        finished).  Both must have completed (result() / drain())."""
        start, _ = self._spans[later]
        _, end = self._spans[earlier]
        return float(start.elapsed_time(end))

    def result(self, ticket: int, numpy_output: bool = True):
        """The audio of one submitted buffer, [C, A, ch] float32 (the shard's block after Tuner.shard)."""
        ev, audio, _ = self._pending.pop(ticket)
        device_output = self._base._cuda and not numpy_output
        if device_output:
            cur = self._torch.cuda.current_stream()
            cur.wait_event(ev)                                 # stream-ordered hand-over, no host wait
            for a in ([b[2] for b in audio] if isinstance(audio, list) else [audio]):
                a.record_stream(cur)
        else:
            ev.synchronize()
        if isinstance(audio, list):                            # submit(each=True): Tuner.run_each's list
            return self._base._each_result(audio, device_output)
        return audio if device_output else hip.to_host(audio)

    def hold_current_stream(self, ticket: int):
        """Order the CURRENT stream behind the last READ of one submitted buffer's input (the wideband FFT of
        Tuner.load; the channel stages read the spectrum, not the input): for a caller that recycles the buffer's
        storage stream-ordered (``with feeder.next() as x: t = lanes.submit(x); lanes.hold_current_stream(t)`` -- the
        Feeder frees the slot behind the current stream, which must therefore wait for the lane that still reads it).
        It does NOT wait for the buffer's audio: the next submit() waits for the current stream, and a wait for the
        whole buffer there would run the lanes one after the other."""
        self._torch.cuda.current_stream().wait_event(self._pending[ticket][2])

    def drain(self):
        """Wait for everything submitted so far (results stay collectable)."""
        for ev, _, _ in self._pending.values():
            ev.synchronize()

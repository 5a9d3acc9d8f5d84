"""Wire format of the reference's multi-channel server.

examples/multi_fm_server.py:103-106 publishes one ZeroMQ multipart message per channel and
buffer: [ int(center_frequency) as 4 little-endian bytes, float32 audio bytes ]; the client
(examples/multi_fm_receiver.py:23-24,47-49) subscribes on that 4-byte prefix.  FM / MFM
payloads are (A, 1) float32, WBFM payloads are (1, A, 2) float32 = interleaved L, R.
These helpers cut a batched [C, A, ch] audio block into those messages and back; the socket
itself (pyzmq) stays outside the package.
"""

import numpy as np

__all__ = ["frames", "parse_frame"]


def frames(channels, audio):
    """[(address_bytes, payload_bytes)] for a [C, A, ch] block, in channel order."""
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    if audio.ndim != 3 or audio.shape[0] != len(channels):
        raise ValueError("audio must be [channels, samples, audio_channels]")
    return [[ch.address_bytes, audio[i].tobytes()] for i, ch in enumerate(channels)]


def parse_frame(message, audio_channels):
    """(center frequency in Hz, float32 [A, audio_channels]) from one multipart message."""
    address, payload = message
    freq = int.from_bytes(address, byteorder="little")
    pcm = np.frombuffer(payload, dtype=np.float32)
    return freq, pcm.reshape(-1, audio_channels)

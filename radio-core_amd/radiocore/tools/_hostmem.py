"""Host staging memory for the ingest side of the path.

The reference allocates `cusignal.get_shared_mem` (CUDA mapped memory) when cuda=True
(radiocore/tools/buffer.py:43, ringbuffer.py:51).  The MI355X equivalent is page-locked
host memory: the 1.92 GB wideband buffer then crosses PCIe by DMA (hipMemcpyAsync from
pinned memory) and can overlap the previous buffer's kernels.  torch supplies the pinned
allocation (memory plumbing only); with cuda=False it is a plain numpy array and no GPU or
torch is needed.
"""

import numpy as np


def zeros(size, dtype, pinned):
    """1-D zero array of `size` elements; page-locked when `pinned`."""
    dt = np.dtype(dtype)
    if not pinned:
        return np.zeros(int(size), dtype=dt), None
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("cuda=True needs a HIP device for page-locked host memory")
    raw = torch.zeros(int(size) * dt.itemsize, dtype=torch.uint8).pin_memory()
    return raw.numpy().view(dt), raw          # keep `raw` alive: it owns the pinned pages

"""Decoded frames -> HBM: pinned double buffer, upload on a side stream, event hand-off.

Real clips are bounded by getting decoded frames to the GPU, not by the kernels: a 720p frame is
2.76 MB over PCIe, and the reference's loop (frameID/data.py:211-230 -> segment_video.py:44)
uploads one pageable float batch at a time, synchronously.  Here the next batch is read from the
decoder straight into page-locked memory and uploaded on its own HIP stream while the GPU
classifies the current one; the classify stream only ever waits on an event.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


class BatchUploader:
    """Yields u8 device batches [k, rows, W, 3] produced by `fill(host_array) -> k` callbacks.

    `fill` writes up to `batch` frames into the page-locked numpy array it is given and returns how
    many it wrote (0 = end of input).  A yielded tensor is valid until the next batch is requested:
    kernels queued on the current stream before then are ordered against the buffer's reuse; do
    not keep the tensor."""

    def __init__(self, device, batch, frame_shape, depth=2):
        self.device = torch.device(device)
        self.batch = int(batch)
        shape = (self.batch,) + tuple(int(v) for v in frame_shape)
        self.depth = max(2, int(depth))
        self.host = [torch.empty(shape, dtype=torch.uint8, pin_memory=True) for _ in range(self.depth)]
        self.host_np = [h.numpy() for h in self.host]
        self.dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(self.depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.uploaded = [torch.cuda.Event() for _ in range(self.depth)]    # the slot's H2D copy is done
        self.consumed = [torch.cuda.Event() for _ in range(self.depth)]    # the consumer's kernels on the slot are done
        self._used = [False] * self.depth

    def _issue(self, slot, fill):
        if self._used[slot]:
            self.uploaded[slot].synchronize()      # the previous upload out of this pinned buffer has finished
        k = int(fill(self.host_np[slot]))
        if k <= 0:
            return 0
        with torch.cuda.stream(self.copy_stream):
            if self._used[slot]:
                self.copy_stream.wait_event(self.consumed[slot])
            self.dev[slot][:k].copy_(self.host[slot][:k], non_blocking=True)
            self.uploaded[slot].record(self.copy_stream)
        self._used[slot] = True
        return k

    def batches(self, fill):
        try:
            slot, k = 0, self._issue(0, fill)
            while k:
                compute = torch.cuda.current_stream(self.device)
                compute.wait_event(self.uploaded[slot])
                yield self.dev[slot][:k]
                # whatever the consumer queued on its stream for this batch is now in front of this event
                self.consumed[slot].record(torch.cuda.current_stream(self.device))
                # read + upload the next batch while the GPU works on the one just handed over
                nxt = (slot + 1) % self.depth
                k = self._issue(nxt, fill)
                slot = nxt
        finally:
            # A consumer that stops early (a frame limit, an exception) may leave an upload in flight.
            # The device buffers belong to the allocating stream's pool: once this object is dropped
            # they can be handed to other tensors, so no copy may still be writing them.
            self.copy_stream.synchronize()


class ParallelCopier:
    """Copies frames (optionally only some rows of each) from a decoded-frame array into a batch
    buffer with a few threads: numpy releases the interpreter lock inside the copies, and one core
    moves only a few GB/s."""

    def __init__(self, workers=None):
        if workers is None:
            workers = int(os.environ.get("CUTDET_COPY_THREADS", "0")) or min(8, os.cpu_count() or 1)
        self.workers = max(1, int(workers))
        self._pool = ThreadPoolExecutor(self.workers) if self.workers > 1 else None

    def copy(self, src, out, rows=None):
        """out[i] = src[i] (or src[i][rows]) for i < len(src)."""
        n = src.shape[0]
        idx = None if rows is None else np.asarray(rows, dtype=np.intp)

        def work(a, b):
            for i in range(a, b):
                if idx is None:
                    np.copyto(out[i], src[i])
                else:
                    np.take(src[i], idx, axis=0, out=out[i])

        if self._pool is None or n < 2 * self.workers:
            work(0, n)
            return
        step = -(-n // self.workers)
        list(self._pool.map(lambda a: work(a, min(n, a + step)), range(0, n, step)))

"""Per-frame scores -> shot segments, with the reference's public surface.

Mirrors `Segmentation` of reference frameID/segmentation.py:26-196 (constructor from a
[N,3] score tensor, `.te`, `len()`, `glue_orphans`, `combine_adjacent_segments`,
`write_csv`) on top of the native library:

  * the per-frame max/argmax and the run-length encoding (reference :37-60) run on the GPU
    (cd_max_argmax + cd_rle);
  * orphan gluing and adjacent merging (reference :91-183) are sequential and O(#segments);
    they run on the host in cd_glue_segments;
  * `.te` holds the same five CPU tensors with the same dtypes (int64 x4, float32).

A score tensor with no rows, and a table that collapses to a single segment that is still
an orphan, raise IndexError exactly where the reference does.
"""
import csv
import ctypes as C

import numpy as np
import torch

from cutdet import _lib, api
from cutdet.pipeline import DEFAULT_COLUMNS, KEEP_LONE_ORPHAN, csv_rows

from .data import SupervisedFrameDataset

_TYPE_MAP = SupervisedFrameDataset.lab_enum
_INVERSE_TYPE_MAP = {v: k for k, v in _TYPE_MAP.items()}

_KEYS = ("end_frames", "frame_types", "run_lengths", "start_frames", "score_means")
GLUE, MERGE = 1, 2


def _host_pass(te, real_threshold, blank_threshold, steps):
    """Run cd_glue_segments over a `te` dict; returns the new dict."""
    arrs = {k: np.ascontiguousarray(te[k].numpy()).copy() for k in _KEYS}
    n = C.c_int64(arrs["end_frames"].shape[0])
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = _lib.lib.cd_glue_segments(p(arrs["end_frames"]), p(arrs["frame_types"]), p(arrs["run_lengths"]),
                                   p(arrs["start_frames"]), p(arrs["score_means"]), C.byref(n),
                                   int(real_threshold), int(blank_threshold), steps)
    if rc != _lib.CD_OK:
        msg = (_lib.lib.cd_last_error() or b"").decode()
        if "index out of bounds" in msg:
            raise IndexError("index 1 is out of bounds for dimension 0 with size 1")
        raise _lib.CutdetError(rc, msg)
    return {k: torch.from_numpy(arrs[k][: n.value].copy()) for k in _KEYS}


class Segmentation:
    """Run-length encodes per-frame scores and repairs short segments."""

    def __init__(self, scores, frame_offset=0):
        if not torch.is_tensor(scores):
            scores = torch.as_tensor(np.asarray(scores, dtype=np.float32))
        if scores.dim() != 2:
            raise ValueError("scores must be [N, classes]")
        if scores.shape[0] == 0:
            raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
        eng = api.default_engine(scores.device if scores.is_cuda else "cuda:0")
        dev_scores = scores.to(device=eng.device, dtype=torch.float32)
        highest, classes = eng.max_argmax(dev_scores)
        dev = eng.rle(classes, highest, frame_offset=frame_offset)
        self.te = {k: dev[k].cpu() for k in _KEYS}

    @classmethod
    def from_stream(cls, classes, highest, frame_offset=0):
        """From the argmax / max streams the forward pass already produced on the GPU."""
        self = cls.__new__(cls)
        eng = api.default_engine(classes.device)
        dev = eng.rle(classes, highest, frame_offset=frame_offset)
        self.te = {k: dev[k].cpu() for k in _KEYS}
        return self

    def __len__(self):
        return self.te["end_frames"].shape[0]

    def glue_orphans(self, real_threshold=100, blank_threshold=10, keep_lone_orphan=False):
        """Glue short segments into a neighbour, least confident first (reference :91-166).  A clip
        that collapses to one segment that is still short raises IndexError, as the reference does
        (:110-116 index row 1 of a one-row table); `keep_lone_orphan=True` keeps that segment."""
        self.te = _host_pass(self.te, real_threshold, blank_threshold,
                             GLUE | (KEEP_LONE_ORPHAN if keep_lone_orphan else 0))

    def combine_adjacent_segments(self):
        """Combine adjacent segments of the same type."""
        self.te = _host_pass(self.te, 0, 0, MERGE)

    def write_csv(self, file_path, columns=DEFAULT_COLUMNS):
        """`start_frame,type` rows, CRLF line ends, no header (reference :185-196).  `columns` may add
        `end_frame`, `run_length` and `score_mean` (see cutdet.pipeline.COLUMNS)."""
        table = {k: v.numpy() for k, v in self.te.items()}
        with open(file_path, "w", newline="") as f:
            csv.writer(f, delimiter=",").writerows(csv_rows(table, columns))

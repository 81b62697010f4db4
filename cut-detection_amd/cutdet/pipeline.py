"""Clip-level pipeline: u8 frames resident on the GPU -> per-frame streams -> segments -> CSV.

This is the data path of the reference's driver loop (segment_video.py:38-77) with the tensor
work moved into the native library: batches of decoded frames go through the fused
resize+classify call, the argmax/max streams stay on the GPU in clip-long buffers, and one
run-length pass plus the host-side repair turns them into the CSV rows.

With several ranks (one process per GPU, frames sharded by contiguous range) the only
exchange is one all-gather of the per-frame (class, max) streams -- 12 bytes per frame -- after
which rank 0 segments the whole clip exactly as a single GPU would.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

CLASS_NAMES = ("a22", "ez", "b")   # ids 0, 1, 2 (reference frameID/data.py:116)
TABLE_KEYS = ("end_frames", "frame_types", "run_lengths", "start_frames", "score_means")
GLUE, MERGE, KEEP_LONE_ORPHAN = 1, 2, 4     # `steps` bits of cd_glue_segments


def glue_and_merge(table, real_threshold=100, blank_threshold=10, steps=3):
    """Host pass over a segment table {key: numpy array}; returns a new table.
    steps: bit 0 glue orphans (reference segmentation.py:91-166), bit 1 merge equal
    neighbours (:168-183), bit 2 (KEEP_LONE_ORPHAN) keep a clip that collapses to one short
    segment.  Without bit 2 it raises IndexError where the reference does."""
    a = {k: np.ascontiguousarray(table[k]).copy() for k in TABLE_KEYS}
    n = C.c_int64(a["end_frames"].shape[0])
    p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = _lib.lib.cd_glue_segments(p(a["end_frames"]), p(a["frame_types"]), p(a["run_lengths"]),
                                   p(a["start_frames"]), p(a["score_means"]), C.byref(n),
                                   int(real_threshold), int(blank_threshold), int(steps))
    if rc != _lib.CD_OK:
        msg = (_lib.lib.cd_last_error() or b"").decode()
        if "index out of bounds" in msg:
            raise IndexError("index 1 is out of bounds for dimension 0 with size 1")
        raise _lib.CutdetError(rc, msg)
    return {k: a[k][: n.value] for k in TABLE_KEYS}


# CSV columns: the reference writes the first two and nothing else (segmentation.py:185-196)
DEFAULT_COLUMNS = ("start", "type")
COLUMNS = ("start", "type", "end_frame", "run_length", "score_mean")


def check_columns(columns):
    columns = tuple(columns)
    bad = [c for c in columns if c not in COLUMNS]
    if bad or not columns:
        raise ValueError(f"unknown CSV column(s) {bad}; choose from {COLUMNS}")
    return columns


def csv_rows(table, columns=DEFAULT_COLUMNS):
    """Rows of a segment table as tuples of str, one field per requested column.  `score_mean`
    is written with 9 significant digits (enough to restore the float32)."""
    columns = check_columns(columns)
    src = {"start": ("start_frames", lambda v: "%d" % int(v)),
           "type": ("frame_types", lambda v: CLASS_NAMES[int(v)]),
           "end_frame": ("end_frames", lambda v: "%d" % int(v)),
           "run_length": ("run_lengths", lambda v: "%d" % int(v)),
           "score_mean": ("score_means", lambda v: "%.9g" % float(v))}
    cols = [[fmt(v) for v in table[key]] for key, fmt in (src[c] for c in columns)]
    return list(zip(*cols))


def csv_bytes(table, columns=DEFAULT_COLUMNS):
    """CSV text of a segment table: CRLF line ends, no header.  The default columns give the
    reference's file byte for byte (`start,type`, segmentation.py:185-196)."""
    return b"".join((",".join(r) + "\r\n").encode() for r in csv_rows(table, columns))


class SegmentCsvStream:
    """One CSV for many clips, written as the clips finish (the multi-clip throughput mode: every
    GPU segments its own clips and appends to a file of its own or, through a lock held by the
    caller, to a shared one).  Rows are `clip,<columns...>`; each clip's rows are written and
    flushed in one piece, so a reader following the file never sees half a clip."""

    def __init__(self, path_or_file, columns=DEFAULT_COLUMNS, clip_column=True, lock=False):
        """`lock`: take an exclusive advisory lock on the file (flock) around each clip's write, so
        that several processes - one per GPU, each with the file open for appending - can share it."""
        self.columns = check_columns(columns)
        self.clip_column = bool(clip_column)
        self.lock = bool(lock)
        self._own = isinstance(path_or_file, (str, bytes)) or hasattr(path_or_file, "__fspath__")
        self._f = open(path_or_file, "ab") if self._own else path_or_file
        self.clips = 0
        self.rows = 0

    def write_clip(self, clip_id, table):
        rows = csv_rows(table, self.columns)
        if self.clip_column:
            cid = str(clip_id)
            if any(ch in cid for ch in ',"\r\n'):
                cid = '"' + cid.replace('"', '""') + '"'
            rows = [(cid,) + r for r in rows]
        data = b"".join((",".join(r) + "\r\n").encode() for r in rows)
        if self.lock:
            import fcntl
            fcntl.flock(self._f.fileno(), fcntl.LOCK_EX)
            try:
                self._f.write(data)
                self._f.flush()
            finally:
                fcntl.flock(self._f.fileno(), fcntl.LOCK_UN)
        else:
            self._f.write(data)
            self._f.flush()
        self.clips += 1
        self.rows += len(rows)

    def close(self):
        if self._own and self._f is not None:
            self._f.close()
        self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def shard_range(total, rank, world):
    """Frame range [lo, hi) of `rank` when `total` frames are split into `world` contiguous ranges of
    ceil(total / world) frames (the last ranges may be short or empty)."""
    per = -(-int(total) // int(world))
    return min(total, rank * per), min(total, (rank + 1) * per)


def gather_streams(cls, maxv, group=None, total=None, force_collective=False):
    """(cls, maxv) of the whole clip on every rank, from each rank's contiguous frame range, in ONE
    all-gather.  Ranks own the ranges of shard_range(total, rank, world), so every count follows from
    `total` (required with more than one rank; all ranks must pass the same value - it sizes the
    collective).  Each rank contributes one fixed-size payload: a 16-byte header (the number of frames
    it actually holds, and `total`), [cap] int64 classes, [cap] float32 maxima, cap = ceil(total /
    world).  A rank whose count is not what the split gives it still sends a payload of the right size,
    so the collective completes, and then EVERY rank raises ValueError from the gathered headers (no
    rank is left waiting inside the collective).  Works on any backend (RCCL for GPU tensors, gloo
    for CPU tensors); without an initialised process group it is the identity.  `force_collective`
    runs the all-gather even in a one-rank group (to exercise the backend)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return cls, maxv
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return cls, maxv
    rank = dist.get_rank(group)
    n_local = int(cls.shape[0])
    if total is None:
        if world > 1:
            raise ValueError("gather_streams: `total` (frames of the whole clip) is required with more than one rank")
        total = n_local
    total = int(total)
    counts = [hi - lo for lo, hi in (shard_range(total, r, world) for r in range(world))]
    if maxv.shape[0] != n_local:
        raise ValueError(f"rank {rank}: {n_local} classes but {maxv.shape[0]} maxima")
    out_dev = cls.device
    # gloo moves host memory only: stage through the CPU when the group is not RCCL
    dev = out_dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    cap = (max(counts) + 1) // 2 * 2          # even, so every rank's slot stays 8-byte aligned
    slot = 16 + cap * 12
    k = min(n_local, cap)
    pay = torch.zeros((slot,), dtype=torch.uint8, device=dev)
    pay[:16] = torch.tensor([n_local, total], dtype=torch.int64).view(torch.uint8).to(dev)
    pay[16: 16 + k * 8] = cls[:k].to(dev).contiguous().view(torch.uint8)
    pay[16 + cap * 8: 16 + cap * 8 + k * 4] = maxv[:k].to(dev).contiguous().view(torch.uint8)
    allpay = torch.empty((world * slot,), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allpay, pay, group=group)
    allpay = allpay.view(world, slot)
    heads = allpay[:, :16].contiguous().view(torch.int64).cpu().tolist()
    bad = [(r, h[0], counts[r]) for r, h in enumerate(heads) if h[0] != counts[r] or h[1] != total]
    if bad:
        r, got, want = bad[0]
        raise ValueError(f"rank {r} holds {got} frames; the contiguous split of {total} frames over "
                         f"{world} ranks gives it {want} (seen by rank {rank})")
    out_cls = torch.cat([allpay[r, 16: 16 + counts[r] * 8].view(torch.int64) for r in range(world)])
    out_max = torch.cat([allpay[r, 16 + cap * 8: 16 + cap * 8 + counts[r] * 4].view(torch.float32) for r in range(world)])
    return out_cls.to(out_dev), out_max.to(out_dev)


def finish_table(initial, real_threshold=100, blank_threshold=10, keep_lone_orphan=False, columns=DEFAULT_COLUMNS):
    """Run-length table of the whole clip -> {"initial", "glued", "merged": tables, "csv": bytes}
    (reference segment_video.py:62-77: glue_orphans, combine_adjacent_segments, write_csv)."""
    glued = glue_and_merge(initial, real_threshold, blank_threshold,
                           steps=GLUE | (KEEP_LONE_ORPHAN if keep_lone_orphan else 0))
    merged = glue_and_merge(glued, real_threshold, blank_threshold, steps=MERGE)
    return {"initial": initial, "glued": glued, "merged": merged, "csv": csv_bytes(merged, columns)}


class ClipClassifier:
    """Classifies a clip batch by batch into clip-long device buffers, then segments it."""

    def __init__(self, engine, n_frames):
        self.engine = engine
        self.n_frames = int(n_frames)
        dev = engine.device
        self.logits = torch.empty((self.n_frames, 3), dtype=torch.float32, device=dev)
        self.maxv = torch.empty((self.n_frames,), dtype=torch.float32, device=dev)
        self.cls = torch.empty((self.n_frames,), dtype=torch.int64, device=dev)

    def classify(self, frames_u8, offset, chan_order=_lib.CD_BGR):
        """frames_u8: u8 [b,H0,W0,3] on the GPU; results land at rows offset .. offset+b."""
        b = frames_u8.shape[0]
        sl = slice(offset, offset + b)
        self.engine.forward_u8(frames_u8, chan_order=chan_order,
                               out=(self.logits[sl], self.maxv[sl], self.cls[sl]))

    def classify_range(self, frames_u8, offset=0, batch=256, chan_order=_lib.CD_BGR, overlap=True):
        """Classifies frames_u8 (u8 [N,H0,W0,3] on the GPU) in batches; results land at rows
        offset .. offset+N.  With `overlap` (default) the whole range is ONE library call
        (cd_forward_u8_range): consecutive batches overlap on two streams inside the library, and a
        geometry that needs a real resize is resized ahead on a third stream.  Without it every batch
        is one fused cd_forward_u8 call on the current stream.  The results are bit-identical."""
        n = frames_u8.shape[0]
        if n == 0:
            return
        if overlap:
            sl = slice(offset, offset + n)
            self.engine.forward_u8_range(frames_u8, batch, (self.logits[sl], self.maxv[sl], self.cls[sl]),
                                         chan_order=chan_order)
            return
        for a in range(0, n, batch):
            self.classify(frames_u8[a:a + batch], offset + a, chan_order)

    def gathered_streams(self, n_local, group=None, total=None):
        return gather_streams(self.cls[:n_local], self.maxv[:n_local], group, total)

    def segment(self, n_local=None, real_threshold=100, blank_threshold=10, group=None, frame_offset=0,
                total=None):
        """Segments the clip: with a process group, this rank's first n_local frames are its range of
        the `total`-frame clip (see gather_streams) and every rank gets the whole clip's result.
        Returns {"initial", "glued", "merged": tables, "csv": bytes}."""
        n_local = self.n_frames if n_local is None else int(n_local)
        cls, mx = self.gathered_streams(n_local, group, total)
        dev = self.engine.rle(cls, mx, frame_offset=frame_offset)
        return finish_table({k: dev[k].cpu().numpy() for k in TABLE_KEYS}, real_threshold, blank_threshold)

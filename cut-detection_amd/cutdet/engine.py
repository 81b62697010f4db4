"""Torch-facing wrapper of the native handle: tensors in, tensors out, nothing computed here.

PyTorch supplies device memory and the current HIP stream; every operation is one call into
libcutdet.so.  Inputs must live on the engine's GPU -- there is no CPU path.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check

NET_W = 256


def output_size(height, width, resize=NET_W):
    """(new_height, new_width) with the reference's rounding (frameID/data.py:199-202)."""
    return int(height * (resize / width)), resize


def sampled_source_rows(height, width, resize=NET_W):
    """Source rows the resize reads when its vertical pass is a pure point sample (the source height
    an odd integer multiple of the output height, e.g. 720 -> 144 reads rows 5y+2), else None.
    Uploading only these rows - 1/5 of a 720p frame - and classifying them with
    forward_u8(..., net_h=len(rows)) gives bit-identical results: the vertical scale is then 1."""
    nh, _ = output_size(height, width, resize)
    if nh < 1 or height % nh or (height // nh) % 2 == 0:
        return None
    s = height // nh
    return [s * y + (s - 1) // 2 for y in range(nh)]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    """One classifier instance on one GPU."""

    def __init__(self, blob, device="cuda:0", dtype="f32"):
        """dtype "f32": the reference's arithmetic.  "bf16_split": every convolution on the bf16 matrix
        cores with each fp32 value carried as hi + lo bfloat16 and fp32 accumulation (logits within 1e-3
        of f32; measured a few 1e-5).  "bf16": plain bfloat16 operands (logits move by a few 1e-2).
        Outputs are fp32 tensors in all three."""
        if dtype not in _lib.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_lib.DTYPES)}, got {dtype!r}")
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("cutdet.Engine runs on an MI355X only; got device %r (no CPU fallback)" % (device,))
        if not torch.cuda.is_available():
            raise RuntimeError("cutdet.Engine: no GPU visible (torch.cuda.is_available() is False)")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self._blob = bytes(blob)
        self._h = C.c_void_p(0)
        check(lib.cd_create(C.byref(self._h), self._blob, len(self._blob), index, _lib.DTYPES[dtype]))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.cd_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_dev(self, t, name, dtype):
        if not torch.is_tensor(t) or t.device != self.device:
            raise RuntimeError(f"{name} must be a tensor on {self.device} (got {getattr(t, 'device', type(t))})")
        if t.dtype != dtype:
            raise TypeError(f"{name} must be {dtype}, got {t.dtype}")

    def _frame_layout(self, frames):
        self._check_dev(frames, "frames", torch.uint8)
        if frames.dim() != 4 or frames.shape[3] != 3 or frames.stride(3) != 1 or frames.stride(2) != 3:
            raise ValueError("frames must be u8 [n, H0, W0, 3] with interleaved pixels")
        n, H0, W0, _ = frames.shape
        return n, H0, W0, frames.stride(1), frames.stride(0) if n > 1 else max(frames.stride(0), frames.stride(1) * H0)

    def set_option(self, name, value):
        """cd_set_option: how forward_u8_range schedules its work ("range_by_stage", "fuse_blocks12",
        "range_one_stream") and which of two kernels runs ("split_two_waves": block 2 of bf16_split;
        "resize_rows": the stand-alone resize); value 0 or 1.  Every setting gives the same bits."""
        check(lib.cd_set_option(self._h, _lib.OPTIONS[name], int(bool(value))))

    def reserve(self, max_frames, net_h=144):
        check(lib.cd_reserve(self._h, int(max_frames), int(net_h)))

    # ------------------------------------------------------------------ operations
    def preprocess(self, frames, chan_order=_lib.CD_BGR, resize=NET_W, out=None):
        """u8 [n,H0,W0,3] -> f32 [n,3,h,resize] RGB/255 (reference frameID/data.py:218-228)."""
        n, H0, W0, rs, fs = self._frame_layout(frames)
        nh, nw = output_size(H0, W0, resize)
        if out is None:
            out = torch.empty((n, 3, nh, nw), dtype=torch.float32, device=self.device)
        check(lib.cd_preprocess(self._h, _ptr(frames), n, H0, W0, rs, fs, chan_order, _ptr(out), nh, nw,
                                self._stream()))
        return out

    def resize_u8(self, frames, resize=NET_W, out=None):
        """u8 [n,H0,W0,3] -> u8 [n,h,resize,3]: cv2.resize INTER_LINEAR alone (reference
        frameID/data.py:220-222), channel order unchanged."""
        n, H0, W0, rs, fs = self._frame_layout(frames)
        nh, nw = output_size(H0, W0, resize)
        if out is None:
            out = torch.empty((n, nh, nw, 3), dtype=torch.uint8, device=self.device)
        else:
            self._check_dev(out, "out", torch.uint8)
            if tuple(out.shape) != (n, nh, nw, 3) or not out.is_contiguous():
                raise ValueError(f"out must be a contiguous u8 tensor of shape {(n, nh, nw, 3)}")
        if n:
            check(lib.cd_resize_u8(self._h, _ptr(frames), n, H0, W0, rs, fs, _ptr(out), nh, nw, self._stream()))
        return out

    def nv12_to_bgr(self, y, uv, y_rows=None, uv_rows=None, chan_order=_lib.CD_BGR, out=None):
        """NV12 planes on the GPU - y u8 [n, Hy, W], uv u8 [n, Huv, W] (views with padded rows are
        fine) - to interleaved pixels u8 [n, rows, W, 3].  Output row r uses luma row y_rows[r] and
        chroma row uv_rows[r] (int32 device tensors; default: every luma row and row r // 2)."""
        self._check_dev(y, "y", torch.uint8)
        self._check_dev(uv, "uv", torch.uint8)
        if y.dim() != 3 or uv.dim() != 3 or y.shape[0] != uv.shape[0] or y.shape[2] != uv.shape[2] or \
                y.stride(2) != 1 or uv.stride(2) != 1:
            raise ValueError("y and uv must be u8 [n, rows, W] with contiguous rows")
        n, hy, w = y.shape
        for name, t in (("y_rows", y_rows), ("uv_rows", uv_rows)):
            if t is not None:
                self._check_dev(t, name, torch.int32)
        if (y_rows is None) != (uv_rows is None) and y_rows is None:
            raise ValueError("uv_rows without y_rows")
        n_rows = hy if y_rows is None else int(y_rows.shape[0])
        if uv_rows is not None and uv_rows.shape[0] != n_rows:
            raise ValueError("y_rows and uv_rows differ in length")
        if y_rows is None and uv.shape[1] * 2 < hy:
            raise ValueError("uv plane too short for the luma plane")
        if out is None:
            out = torch.empty((n, n_rows, w, 3), dtype=torch.uint8, device=self.device)
        fs = lambda t: t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.stride(1) * t.shape[1])  # noqa: E731
        if n:
            check(lib.cd_nv12_to_bgr(self._h, _ptr(y), _ptr(uv), n, w, y.stride(1), uv.stride(1), fs(y), fs(uv),
                                     _ptr(y_rows.contiguous() if y_rows is not None else None),
                                     _ptr(uv_rows.contiguous() if uv_rows is not None else None), n_rows, _ptr(out),
                                     chan_order, self._stream()))
        return out

    def _outputs(self, n, want_max, out=None):
        if out is not None:
            logits, mx, am = out
            self._check_dev(logits, "out[0]", torch.float32)
            self._check_dev(mx, "out[1]", torch.float32)
            self._check_dev(am, "out[2]", torch.int64)
            if tuple(logits.shape) != (n, 3) or tuple(mx.shape) != (n,) or tuple(am.shape) != (n,) or not (
                    logits.is_contiguous() and mx.is_contiguous() and am.is_contiguous()):
                raise ValueError("out must be contiguous tensors of shapes (n,3), (n,), (n,)")
            return logits, mx, am
        logits = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        if not want_max:
            return logits, None, None
        return (logits, torch.empty((n,), dtype=torch.float32, device=self.device),
                torch.empty((n,), dtype=torch.int64, device=self.device))

    def forward(self, x, want_max=False):
        """f32 [n,3,h,256] -> logits [n,3] (and max, argmax)."""
        self._check_dev(x, "x", torch.float32)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"x must be [n,3,h,{NET_W}], got {tuple(x.shape)}")
        x = x.contiguous()
        n, _, nh, nw = x.shape
        logits, mx, am = self._outputs(n, want_max)
        if n:
            check(lib.cd_forward(self._h, _ptr(x), n, nh, nw, _ptr(logits), _ptr(mx), _ptr(am), self._stream()))
        return (logits, mx, am) if want_max else logits

    def forward_u8(self, frames, chan_order=_lib.CD_BGR, resize=NET_W, want_max=True, out=None, net_h=None):
        """u8 [n,H0,W0,3] -> logits (and max, argmax) with the resize fused into the first layer.
        `out` = (logits, max, argmax) tensors to write into (e.g. slices of clip-long buffers).
        `net_h` overrides the output height the reference's size rule would give; it is for frames
        that were already reduced to their sampled rows (see sampled_source_rows)."""
        n, H0, W0, rs, fs = self._frame_layout(frames)
        nh, nw = output_size(H0, W0, resize)
        if net_h is not None:
            nh = int(net_h)
        logits, mx, am = self._outputs(n, want_max, out)
        check(lib.cd_forward_u8(self._h, _ptr(frames), n, H0, W0, rs, fs, chan_order, nh, nw, _ptr(logits),
                                _ptr(mx), _ptr(am), self._stream()))
        return (logits, mx, am) if want_max else logits

    def forward_u8_range(self, frames, batch, out, chan_order=_lib.CD_BGR, resize=NET_W, net_h=None):
        """The whole hot loop in one call: u8 [N,H0,W0,3] frames resident on the GPU, classified in
        batches of `batch`; `out` = (logits [N,3], max [N], argmax [N]) to write into.  Bit-identical to
        forward_u8 batch by batch; consecutive batches overlap on two streams inside the library and a
        geometry that needs a real resize is resized one batch ahead on a third (cd_forward_u8_range)."""
        n, H0, W0, rs, fs = self._frame_layout(frames)
        nh, nw = output_size(H0, W0, resize)
        if net_h is not None:
            nh = int(net_h)
        logits, mx, am = self._outputs(n, True, out)
        if n:
            check(lib.cd_forward_u8_range(self._h, _ptr(frames), n, int(batch), H0, W0, rs, fs, chan_order, nh, nw,
                                          _ptr(logits), _ptr(mx), _ptr(am), self._stream()))
        return logits, mx, am

    def forward_u8_static(self, frames, out, chan_order=_lib.CD_BGR, resize=NET_W, net_h=None):
        """forward_u8 through the library's entry point that allocates nothing: `out` = (logits, max,
        argmax) must be given, and the batch must fit the library's fixed activation array (34 frames
        at 144 input rows; CutdetError CD_ECAP beyond).  Calls that overlap on one GPU race."""
        n, H0, W0, rs, fs = self._frame_layout(frames)
        nh, nw = output_size(H0, W0, resize)
        if net_h is not None:
            nh = int(net_h)
        logits, mx, am = self._outputs(n, True, out)
        check(lib.cd_forward_u8_static(self._h, _ptr(frames), n, H0, W0, rs, fs, chan_order, nh, nw, _ptr(logits),
                                       _ptr(mx), _ptr(am), self._stream()))
        return logits, mx, am

    def max_argmax(self, scores):
        """f32 [n,k] -> (max f32 [n], argmax i64 [n]); torch.max(scores, 1) semantics."""
        self._check_dev(scores, "scores", torch.float32)
        if scores.dim() != 2 or scores.shape[1] < 1:
            raise ValueError("scores must be [n, k]")
        scores = scores.contiguous()
        n, k = scores.shape
        mx = torch.empty((n,), dtype=torch.float32, device=self.device)
        am = torch.empty((n,), dtype=torch.int64, device=self.device)
        if n:
            check(lib.cd_max_argmax(self._h, _ptr(scores), n, k, _ptr(mx), _ptr(am), self._stream()))
        return mx, am

    def eval_accumulate(self, logits, labels, correct, total, loss_sum):
        """Adds one labelled batch to the validation accumulators (device tensors: correct, total i64
        [classes], loss_sum f64 [1]); see cd_eval_accumulate."""
        self._check_dev(logits, "logits", torch.float32)
        self._check_dev(labels, "labels", torch.int64)
        self._check_dev(correct, "correct", torch.int64)
        self._check_dev(total, "total", torch.int64)
        self._check_dev(loss_sum, "loss_sum", torch.float64)
        if logits.dim() != 2 or labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
            raise ValueError("logits must be [n, k] and labels [n]")
        k = logits.shape[1]
        if correct.numel() != k or total.numel() != k or loss_sum.numel() != 1:
            raise ValueError("accumulators must hold one count per class and one loss sum")
        logits, labels = logits.contiguous(), labels.contiguous()
        if logits.shape[0]:
            check(lib.cd_eval_accumulate(self._h, _ptr(logits), _ptr(labels), logits.shape[0], k, _ptr(correct),
                                         _ptr(total), _ptr(loss_sum), self._stream()))

    def rle(self, cls, maxv, frame_offset=0, cap=None):
        """Class stream -> dict of device tensors (end_frames, frame_types, run_lengths, start_frames,
        score_means, score_sums) trimmed to the number of segments.  Synchronises to read the count."""
        self._check_dev(cls, "cls", torch.int64)
        self._check_dev(maxv, "maxv", torch.float32)
        cls, maxv = cls.contiguous(), maxv.contiguous()
        n = cls.shape[0]
        if n == 0:
            raise IndexError("no frames to segment")
        cap = n if cap is None else int(cap)
        i64 = dict(dtype=torch.int64, device=self.device)
        end, typ, rl, st = (torch.empty((cap,), **i64) for _ in range(4))
        mean = torch.empty((cap,), dtype=torch.float32, device=self.device)
        ssum = torch.empty((cap,), dtype=torch.float64, device=self.device)
        nseg = torch.zeros((1,), **i64)
        check(lib.cd_rle(self._h, _ptr(cls), _ptr(maxv), n, int(frame_offset), _ptr(end), _ptr(typ), _ptr(rl),
                         _ptr(st), _ptr(mean), _ptr(ssum), cap, _ptr(nseg), self._stream()))
        s = int(nseg.item())
        if s > cap:
            raise _lib.CutdetError(_lib.CD_ECAP, f"{s} segments exceed capacity {cap}")
        return {"end_frames": end[:s], "frame_types": typ[:s], "run_lengths": rl[:s], "start_frames": st[:s],
                "score_means": mean[:s], "score_sums": ssum[:s]}

    def timing_enable(self, on=True, stages=None, every=1):
        """Bracket stage launches with HIP events: all stages, or only the named ones, and only every
        `every`-th launch of each (an event pair costs a few microseconds of stream time per launch)."""
        check(lib.cd_timing_interval(self._h, int(every)))
        if on and stages:
            mask = 0
            for name in stages:
                mask |= 1 << (_lib.STAGES.index(name) + 1)
            check(lib.cd_timing_enable(self._h, mask))
        else:
            check(lib.cd_timing_enable(self._h, 1 if on else 0))

    def timing_read(self):
        """{stage: (launches, total_ms)} for the launches bracketed since the last read (syncs)."""
        out = {}
        for i, name in enumerate(_lib.STAGES):
            k, ms = C.c_int64(0), C.c_double(0.0)
            check(lib.cd_timing_read(self._h, i, C.byref(k), C.byref(ms)))
            out[name] = (k.value, ms.value)
        return out

    def synth_frames(self, classes, H0, W0, seed=1234, first_index=0, out=None):
        """Deterministic synthetic clip: u8 [n,H0,W0,3] for per-frame class ids (device u8 tensor)."""
        self._check_dev(classes, "classes", torch.uint8)
        n = classes.shape[0]
        if out is None:
            out = torch.empty((n, H0, W0, 3), dtype=torch.uint8, device=self.device)
        step = 32768
        for a in range(0, n, step):
            b = min(n, a + step)
            check(lib.cd_synth_frames(self._h, _ptr(out[a:b]), b - a, H0, W0, _ptr(classes[a:b].contiguous()),
                                      int(seed) & 0xFFFFFFFF, int(first_index) + a, self._stream()))
        return out

"""Video input with the reference's names (`open_video`, `VideoDataset`), preprocessed on the GPU.

Mirrors reference frameID/data.py:13-31 and :184-234.  Decoding stays outside: a capture
object (OpenCV's when cv2 is importable, else the raw container of cutdet.rawvideo) hands over
u8 BGR frames; resize + float conversion + channel flip + /255 (reference :218-228) happen in
one HIP kernel on batches of frames, and the dataset yields the resulting [3, h, w] float32
tensors -- on the GPU, so the driver's `batch.to(device)` is free.

The training-only datasets of the reference are not part of this package; the class-name table
they define (reference :116) is kept because the segmentation output needs it.
"""
import torch
from torch.utils.data import IterableDataset

import numpy as np

from cutdet import api, ingest, rawvideo
from cutdet.engine import sampled_source_rows


class SupervisedFrameDataset:
    """Holder of the label table only (the reference's training dataset is out of scope)."""

    lab_enum = {"a22": 0, "ez": 1, "b": 2}

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("training datasets are not part of the inference package")


def open_video(video_path):
    """Open a video; returns `(capture, {"fps", "length", "width", "height"})` with int values."""
    if rawvideo.is_raw_video(video_path):
        cap = rawvideo.RawVideoCapture(video_path)
        ids = (rawvideo.CAP_PROP_FPS, rawvideo.CAP_PROP_FRAME_COUNT, rawvideo.CAP_PROP_FRAME_WIDTH,
               rawvideo.CAP_PROP_FRAME_HEIGHT)
    else:
        try:
            import cv2
        except ImportError as e:
            raise ImportError(f"{video_path} is not a CDRAWVID file and OpenCV (cv2) is not installed "
                              "to decode it") from e
        cap = cv2.VideoCapture(video_path)
        ids = (cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_COUNT, cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT)
    fps, length, width, height = (int(cap.get(i)) for i in ids)
    return cap, {"fps": fps, "length": length, "width": width, "height": height}


class VideoDataset(IterableDataset):
    """Iterates the frames of a video as float32 [3, new_height, new_width] RGB tensors in [0,1]."""

    def __init__(self, file_path, resize=None, device="cuda:0", read_ahead=64):
        super().__init__()
        self.cap, self.video_info = open_video(file_path)
        self.device = torch.device(device)
        self.read_ahead = int(read_ahead)
        self.chan_order = getattr(self.cap, "chan_order", 0)
        if resize is not None:
            self.new_width = resize
            self.new_height = int(self.video_info["height"] * (self.new_width / self.video_info["width"]))
        else:
            self.new_width = None
            self.new_height = None
        self._ready = None   # preprocessed frames of the current read-ahead batch
        self._nv12_iter = None
        self._next = 0
        # Rows the resize actually reads when its vertical pass is a point sample (720p: 144 of 720
        # rows): frame_batches() then uploads only those, which is what bounds real clips (2.76 MB per
        # 720p frame over PCIe), and `batch_net_height` tells the classifier that the rows are already
        # the network's.  The raw container never reads the other rows from the file; frames decoded
        # by OpenCV are cut down on the host before the upload.
        self.sampled_rows = None
        if resize is not None:
            self.sampled_rows = sampled_source_rows(self.video_info["height"], self.video_info["width"], resize)
        self.batch_net_height = self.new_height if self.sampled_rows is not None else None
        self._copier = None
        self._ended = False
        self._remaining = None       # frames left of the range set by seek(); None = no limit
        # NV12 containers (what hardware decoders deliver) are uploaded as planes - half the bytes of
        # BGR - and converted on the GPU; sampled-row upload takes the luma rows the resize reads and
        # the chroma row under each of them.
        self.nv12 = bool(getattr(self.cap, "nv12", False))
        if self.nv12:
            self.chan_order = 0          # the converter writes BGR

    # --- decoded frames ------------------------------------------------------------------
    def seek(self, frame_index, max_frames=None):
        """The next frame read is `frame_index` and at most `max_frames` frames follow (None = to the
        end): how a rank that owns the frame range [lo, hi) of a clip opens it, seek(lo, hi - lo).
        OpenCV captures seek through CAP_PROP_POS_FRAMES, as the raw container does."""
        if not self.cap.set(rawvideo.CAP_PROP_POS_FRAMES, int(frame_index)):
            raise RuntimeError(f"the capture cannot seek to frame {frame_index}")
        self._remaining = None if max_frames is None else max(0, int(max_frames))
        self._ended = False
        self._ready = None

    def _fill(self, out, rows):
        """Next decoded frames (only `rows` of each, if given) into out[:k]; returns k."""
        if self._remaining is not None:
            out = out[: min(out.shape[0], self._remaining)]
            if out.shape[0] == 0:
                return 0
        k = self._fill_unbounded(out, rows)
        if self._remaining is not None:
            self._remaining -= k
        return k

    def _fill_unbounded(self, out, rows):
        if hasattr(self.cap, "read_batch_into"):
            if self._copier is None:
                self._copier = ingest.ParallelCopier()
            return self.cap.read_batch_into(out, rows, self._copier)
        idx = None if rows is None else np.asarray(rows, dtype=np.intp)
        k = 0
        while k < out.shape[0] and not self._ended:
            ret, frame = self.cap.read()          # reference frameID/data.py:213
            if not ret:
                self._ended = True                # like the reference's iterator, stop at the first failed read
                break
            if idx is None:
                np.copyto(out[k], frame)
            else:
                np.take(frame, idx, axis=0, out=out[k])
            k += 1
        return k

    def read_u8_batch(self, n, rows=None):
        """Next up-to-n decoded frames as a u8 [k,H,W,3] tensor on the GPU (k = 0 at the end);
        `rows` restricts every frame to those rows.  Synchronous; frame_batches() overlaps."""
        h = self.video_info["height"] if rows is None else len(rows)
        host = np.empty((n, h, self.video_info["width"], 3), dtype=np.uint8)
        k = self._fill(host, rows)
        return torch.from_numpy(host[:k]).to(self.device, non_blocking=False)

    def frame_batches(self, batch_size, sampled=True, overlap=True):
        """Yields u8 GPU batches for the fused classify path (no float frames materialised).  With
        `sampled` and a point-sampling geometry the batches hold only the sampled rows
        ([k, new_height, W, 3]); classify them with net_h=self.batch_net_height.
        With `overlap` (default) batches come through a page-locked double buffer and a copy stream:
        batch i+1 is read and uploaded while the GPU classifies batch i, and a yielded batch is valid
        only until the next one is requested."""
        rows = self.sampled_rows if sampled else None
        self.batch_net_height = self.new_height if rows is not None else None
        if self.nv12:
            yield from self._nv12_batches(batch_size, rows, overlap)
            return
        if not overlap:
            while True:
                b = self.read_u8_batch(batch_size, rows)
                if b.shape[0] == 0:
                    return
                yield b
        h = self.video_info["height"] if rows is None else len(rows)
        up = ingest.BatchUploader(self.device, batch_size, (h, self.video_info["width"], 3))
        yield from up.batches(lambda out: self._fill(out, rows))

    def _nv12_batches(self, batch_size, rows, overlap):
        """NV12 container: plane rows are uploaded ([k, R, W] u8: the luma rows, then the chroma rows)
        and converted to BGR on the GPU (cd_nv12_to_bgr)."""
        H, W = self.video_info["height"], self.video_info["width"]
        eng = api.default_engine(self.device)
        if rows is None:
            file_rows, ny, sel = None, H, (None, None)
        else:
            file_rows = list(rows) + [H + r // 2 for r in rows]          # chroma row r//2 lies under luma row r
            ny = len(rows)
            ident = torch.arange(ny, dtype=torch.int32, device=self.device)
            sel = (ident, ident)                                        # uploaded row r pairs with uploaded chroma row r
        total = H * 3 // 2 if file_rows is None else len(file_rows)

        def convert(planes):
            return eng.nv12_to_bgr(planes[:, :ny], planes[:, ny:], sel[0], sel[1])

        if overlap:
            up = ingest.BatchUploader(self.device, batch_size, (total, W))
            for planes in up.batches(lambda out: self._fill(out, file_rows)):
                yield convert(planes)
            return
        while True:
            host = np.empty((batch_size, total, W), dtype=np.uint8)
            k = self._fill(host, file_rows)
            if k == 0:
                return
            yield convert(torch.from_numpy(host[:k]).to(self.device))

    # --- the reference's iteration protocol ---------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        if self._ready is None or self._next >= self._ready.shape[0]:
            if self.nv12:
                if self._nv12_iter is None:
                    self._nv12_iter = self._nv12_batches(self.read_ahead, None, False)
                raw = next(self._nv12_iter, None)
                if raw is None:
                    raise StopIteration
            else:
                raw = self.read_u8_batch(self.read_ahead)
            if raw.shape[0] == 0:
                raise StopIteration
            eng = api.default_engine(self.device)
            resize = self.new_width if self.new_width is not None else self.video_info["width"]
            self._ready = eng.preprocess(raw, chan_order=self.chan_order, resize=resize)
            self._next = 0
        frame = self._ready[self._next]
        self._next += 1
        return frame

    def __len__(self):
        """Note iterables don't necessarily have len, but this one does."""
        return self.video_info["length"]

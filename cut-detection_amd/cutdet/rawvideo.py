"""A minimal container for already-decoded video: a 64-byte header followed by u8 frames.

Video decoding is outside this package (the reference delegates it to OpenCV,
frameID/data.py:19-24,213); boxes without OpenCV can still feed the pipeline from this format.

    offset  0  8 bytes  magic b"CDRAWVID"
            8  u32 x 6  version (1), width, height, frame count, fps, pixel format
                        (0 BGR, 1 RGB: interleaved 3-byte pixels; 2 NV12: a height x width luma plane
                        followed by a height/2 x width plane of interleaved U,V pairs)
           32  zero padding to 64 bytes
           64  frames, rows top to bottom: height*width*3 bytes each (BGR/RGB) or height*width*3/2 (NV12)
"""
import struct

import numpy as np

MAGIC = b"CDRAWVID"
HEADER_BYTES = 64

# the cv2.CAP_PROP_* ids the reference queries (frameID/data.py:20-24)
CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT = 3, 4, 5, 7
CAP_PROP_POS_FRAMES = 1          # index of the frame the next read() returns (cv2's id for it)


NV12 = 2


def write_raw_video(path, frames, fps=30, chan_order=0):
    """frames: u8 array [n, H, W, 3] (or an iterable of [H, W, 3] arrays); with chan_order=NV12,
    [n, H*3/2, W]: every frame its luma rows followed by its chroma rows."""
    frames = np.asarray(frames, dtype=np.uint8)
    if chan_order == NV12:
        if frames.ndim != 3 or frames.shape[1] % 3 or frames.shape[2] % 2:
            raise ValueError("NV12 frames must be [n, H*3/2, W] uint8 with even H and W")
        n, h, w = frames.shape[0], frames.shape[1] * 2 // 3, frames.shape[2]
    elif frames.ndim != 4 or frames.shape[3] != 3:
        raise ValueError("frames must be [n, H, W, 3] uint8")
    else:
        n, h, w, _ = frames.shape
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<6I", 1, w, h, n, int(fps), int(chan_order)) + b"\0" * (HEADER_BYTES - 32))
        f.write(np.ascontiguousarray(frames).tobytes())


def is_raw_video(path):
    try:
        with open(path, "rb") as f:
            return f.read(8) == MAGIC
    except OSError:
        return False


class RawVideoCapture:
    """The slice of cv2.VideoCapture's interface the pipeline uses, over the raw container."""

    def __init__(self, path):
        with open(path, "rb") as f:
            head = f.read(HEADER_BYTES)
        if len(head) < HEADER_BYTES or head[:8] != MAGIC:
            raise ValueError(f"{path} is not a CDRAWVID file")
        ver, self.width, self.height, self.length, self.fps, self.chan_order = struct.unpack("<6I", head[8:32])
        if ver != 1:
            raise ValueError(f"{path}: unsupported CDRAWVID version {ver}")
        self.nv12 = self.chan_order == NV12
        shape = (self.length, self.height * 3 // 2, self.width) if self.nv12 else (self.length, self.height, self.width, 3)
        self._frames = np.memmap(path, dtype=np.uint8, mode="r", offset=HEADER_BYTES, shape=shape)
        self._pos = 0

    def isOpened(self):
        return True

    def get(self, prop):
        return float({CAP_PROP_FRAME_WIDTH: self.width, CAP_PROP_FRAME_HEIGHT: self.height,
                      CAP_PROP_FPS: self.fps, CAP_PROP_FRAME_COUNT: self.length,
                      CAP_PROP_POS_FRAMES: self._pos}.get(prop, 0))

    def set(self, prop, value):
        """cv2.VideoCapture.set for CAP_PROP_POS_FRAMES: the next read starts at frame `value` (a rank
        that owns a frame range of the clip opens the file and seeks to its first frame).  Returns
        False, like cv2, for a property it cannot set."""
        if prop != CAP_PROP_POS_FRAMES:
            return False
        self._pos = min(max(int(value), 0), self.length)
        return True

    def read(self):
        if self._pos >= self.length:
            return False, None
        frame = np.array(self._frames[self._pos])
        self._pos += 1
        return True, frame

    def read_batch(self, n, rows=None):
        """Up to n consecutive frames as one [k, H, W, 3] array (k = 0 at the end); with `rows`, only
        those rows of each frame ([k, len(rows), W, 3]) - the other rows are never read from the file."""
        a, b = self._pos, min(self.length, self._pos + n)
        self._pos = b
        if rows is None:
            return np.array(self._frames[a:b])
        return np.ascontiguousarray(self._frames[a:b][:, np.asarray(rows, dtype=np.int64)])

    def read_batch_into(self, out, rows=None, copier=None):
        """read_batch() into a caller-owned array (e.g. page-locked memory): fills out[:k] and returns
        k (0 at the end).  `copier`: a cutdet.ingest.ParallelCopier to spread the copy over threads."""
        a, b = self._pos, min(self.length, self._pos + out.shape[0])
        self._pos = b
        if b > a:
            if copier is not None:
                copier.copy(self._frames[a:b], out, rows)
            elif rows is None:
                np.copyto(out[:b - a], self._frames[a:b])
            else:
                np.take(self._frames[a:b], np.asarray(rows, dtype=np.intp), axis=1, out=out[:b - a])
        return b - a

    def release(self):
        self._frames = None

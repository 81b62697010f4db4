"""Oracle side of the synthetic clip generator (test infrastructure).

Integer-only closed form, so the CPU (here, numpy uint32) and the GPU (the library's
generator kernel) regenerate bit-identical u8 frames from (seed, frame index, class).
There is no counterpart in the reference (it ships no data and no tests); the three
frame looks are chosen so that the reference classifier labels them a22 / ez / b with
a wide logit margin.

  a22 (class 0): i.i.d. uniform bytes 0..255
  ez  (class 1): (26,128,26) + 0..7 noise, value-230 horizontal stripes T rows thick every P rows
  b   (class 2): i.i.d. uniform bytes 0..7
"""
import numpy as np

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)


def mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= _M1
    x ^= x >> np.uint32(15)
    x *= _M2
    x ^= x >> np.uint32(16)
    return x


def stripe_geometry(width):
    """(period P, thickness T) in source rows for a `width`-pixel-wide source."""
    return max(1, (12 * width) // 256), (width + 255) // 256


def frame(seed, index, cls, height, width):
    """One u8 frame [height, width, 3] for global frame `index` of class `cls`."""
    with np.errstate(over="ignore"):
        key = mix32(np.uint32(seed) + np.uint32(0x9E3779B9) * np.uint32((index + 1) & 0xFFFFFFFF))
        e = np.arange(height * width * 3, dtype=np.uint32)
        r = mix32(key ^ (e * np.uint32(0x85EBCA6B) + np.uint32(0xC2B2AE35)))
    r = r.reshape(height, width, 3)
    if cls == 0:
        return (r >> np.uint32(24)).astype(np.uint8)
    if cls == 2:
        return (r >> np.uint32(29)).astype(np.uint8)
    if cls != 1:
        raise ValueError("class must be 0, 1 or 2")
    period, thick = stripe_geometry(width)
    base = np.empty((height, width, 3), dtype=np.uint32)
    base[:, :, 0] = 26
    base[:, :, 1] = 128
    base[:, :, 2] = 26
    rows = (np.arange(height) % period) < thick
    base[rows, :, :] = 230
    return (base + (r >> np.uint32(29))).astype(np.uint8)


def clip(seed, classes, height, width, first_index=0):
    """u8 [n, height, width, 3] for per-frame class ids `classes`."""
    classes = np.asarray(classes)
    out = np.empty((classes.shape[0], height, width, 3), dtype=np.uint8)
    for i, c in enumerate(classes):
        out[i] = frame(seed, first_index + i, int(c), height, width)
    return out

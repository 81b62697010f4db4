"""Oracle for frame preprocessing (test infrastructure).  PARITY UNPINNED for the resize.

Restates reference frameID/data.py:197-202 (output size), :218-222 (cv2.resize,
INTER_LINEAR, on the u8 BGR HWC frame) and :225-228 (float32, HWC->CHW, BGR->RGB, /255).

cv2 itself is a third-party dependency that is absent here (opencv-python, unpinned in the
reference's requirements.txt:1), so `resize_linear_u8` follows the published algorithm of
OpenCV's 8-bit INTER_LINEAR resize (imgproc resize.cpp, fixed-point path):
  * per output column  fx = float((dx+0.5)*scale_x - 0.5), sx = floor(fx), fx -= sx;
    sx < 0 -> (sx,fx) = (0,0);  sx >= W-1 -> (sx,fx) = (W-1,0)
  * 11-bit coefficients  a1 = sat_s16(rint(fx*2048)), a0 = sat_s16(rint((1-fx)*2048)) (float math)
  * horizontal pass  S = p[sx]*a0 + p[sx+1]*a1  (int32);  columns whose sx+1 is outside
    the row use S = p[sx]*2048
  * rows: fy/sy the same way but row indices are clamped to [0,H-1] and the
    coefficients are NOT zeroed at the border
  * vertical pass  dst = (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
  * an exact 2x2 downscale is done as a 2x2 box average ((a+b+c+d+2)>>2) instead
Nothing in the reference pins these bytes; the 1280x720 -> 256x144 case reduces to
dst[y,x] = src[5y+2, 5x+2] for every OpenCV version.
"""
import numpy as np

COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS


def output_size(height, width, resize=256):
    """(new_height, new_width) as reference data.py:199-202 (double math, then int())."""
    return int(height * (resize / width)), resize


def _axis_tables(src_n, dst_n, zero_at_border):
    """sample index, i16 coefficient pair per output coordinate; float32/float64 mix as OpenCV."""
    inv = np.float64(dst_n) / np.float64(src_n)
    scale = np.float64(1.0) / inv  # OpenCV forms scale as 1/inv_scale, not src/dst
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    plain = np.ones(dst_n, dtype=bool)  # True: two-tap formula applies
    if zero_at_border:
        lo = s < 0
        f[lo] = 0
        s[lo] = 0
        hi = s >= src_n - 1
        f[hi] = 0
        s[hi] = src_n - 1
        plain = (s + 1) < src_n  # columns at/after xmax use the one-tap form
    c1 = np.rint(f * np.float32(COEF_ONE)).astype(np.int64)
    c0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_ONE)).astype(np.int64)
    c0 = np.clip(c0, -32768, 32767)
    c1 = np.clip(c1, -32768, 32767)
    return s, c0, c1, plain


def resize_linear_u8(src, new_height, new_width):
    """src u8 [H,W,C] -> u8 [new_height,new_width,C], OpenCV 8U INTER_LINEAR semantics."""
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    H, W, _ = src.shape
    if H == 2 * new_height and W == 2 * new_width:
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1, plain = _axis_tables(W, new_width, True)
    sy, b0, b1, _ = _axis_tables(H, new_height, False)
    if (a1 == 0).all() and (b1 == 0).all() and (a0 == COEF_ONE).all() and (b0 == COEF_ONE).all():
        # every tap weight is 1.0: ((2048*((p*2048)>>4))>>16 + 2)>>2 == p, a pure point sample
        return src[np.clip(sy, 0, H - 1)][:, sx]
    p = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, W - 1)
    two = p[:, sx, :] * a0[None, :, None] + p[:, sx1, :] * a1[None, :, None]
    one = p[:, sx, :] * COEF_ONE
    hrow = np.where(plain[None, :, None], two, one)  # [H, new_w, C] int
    r0 = np.clip(sy, 0, H - 1)
    r1 = np.clip(sy + 1, 0, H - 1)
    S0 = hrow[r0]
    S1 = hrow[r1]
    v = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def tensorise(frame_u8, chan_order=0):
    """u8 [h,w,3] -> float32 [3,h,w] RGB in [0,1]; reference data.py:225-228.
    chan_order 0 = BGR in memory (cv2), 1 = already RGB."""
    x = np.asarray(frame_u8).astype(np.float32).transpose(2, 0, 1)
    if chan_order == 0:
        x = x[::-1]
    return (x / np.float32(255.0)).astype(np.float32)


def preprocess(frame_u8, resize=256, chan_order=0):
    """Full P1+P2 for one decoded frame; reference data.py:211-230 minus the decode."""
    h, w, _ = frame_u8.shape
    nh, nw = output_size(h, w, resize)
    return tensorise(resize_linear_u8(frame_u8, nh, nw), chan_order)


def preprocess_batch(frames_u8, resize=256, chan_order=0):
    return np.stack([preprocess(f, resize, chan_order) for f in frames_u8])


# ----------------------------------------------------------------------------------------------
# NV12 -> BGR.  PARITY UNPINNED: the reference never sees NV12 - cv2.VideoCapture.read()
# (frameID/data.py:213) hands it BGR, converted inside OpenCV's bundled decoder.  This restates the
# published fixed-point arithmetic of OpenCV's cvtColor(COLOR_YUV2BGR_NV12) (imgproc color_yuv,
# ITU-R BT.601 limited range, 20-bit coefficients): for a pixel (Y, U, V), U/V shared by a 2x2 block,
#   y = max(0, Y - 16) * 1220542
#   R = sat8((y + 1673527*(V-128)                  + (1 << 19)) >> 20)
#   G = sat8((y -  852492*(V-128) - 409993*(U-128) + (1 << 19)) >> 20)
#   B = sat8((y + 2116026*(U-128)                  + (1 << 19)) >> 20)
NV12_CY, NV12_CVR, NV12_CVG, NV12_CUG, NV12_CUB, NV12_SHIFT = 1220542, 1673527, -852492, -409993, 2116026, 20


def nv12_to_bgr(y_plane, uv_plane, y_rows=None, uv_rows=None, chan_order=0):
    """y_plane u8 [Hy, W], uv_plane u8 [Huv, W] (U, V interleaved) -> u8 [rows, W, 3].
    Output row r takes luma row y_rows[r] (default r) and chroma row uv_rows[r] (default r // 2)."""
    y_plane, uv_plane = np.asarray(y_plane), np.asarray(uv_plane)
    n = y_plane.shape[0] if y_rows is None else len(y_rows)
    yr = np.arange(n) if y_rows is None else np.asarray(y_rows)
    ur = yr // 2 if uv_rows is None else np.asarray(uv_rows)
    Y = y_plane[yr].astype(np.int64)
    uv = uv_plane[ur].astype(np.int64)
    U = np.repeat(uv[:, 0::2], 2, axis=1)[:, :Y.shape[1]] - 128
    V = np.repeat(uv[:, 1::2], 2, axis=1)[:, :Y.shape[1]] - 128
    y = np.maximum(0, Y - 16) * NV12_CY
    half = 1 << (NV12_SHIFT - 1)
    r = np.clip((y + NV12_CVR * V + half) >> NV12_SHIFT, 0, 255)
    g = np.clip((y + NV12_CVG * V + NV12_CUG * U + half) >> NV12_SHIFT, 0, 255)
    b = np.clip((y + NV12_CUB * U + half) >> NV12_SHIFT, 0, 255)
    out = np.stack([b, g, r] if chan_order == 0 else [r, g, b], axis=-1)
    return out.astype(np.uint8)

// Device-side 8-bit bilinear sampling shared by the stand-alone preprocess kernel and the
// fused first-layer kernel.  Arithmetic follows OpenCV's 8U INTER_LINEAR fixed-point path
// (11-bit coefficients; horizontal pass in int32, vertical pass
// (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2), which is what cv2.resize computes at
// reference frameID/data.py:220-222.
#pragma once
#include "cd_internal.h"

namespace cd {

struct SampleArgs {
    const int* sx;      // [w]  left source column (already clamped to [0, W0-1])
    const short* ax;    // [2w] horizontal coefficient pair
    const int* sy;      // [h]  top source row (may be -1 or H0-1; rows are clamped on use)
    const short* by;    // [2h] vertical coefficient pair
    int mode;           // 0 bilinear, 1 point sample, 2 exact 2x2 box average
    int H0, W0;
    int64_t row_stride; // bytes
    // point-sample mode only: when both index tables are affine (sx[x] = xm*x + xa, sy[y] = ym*y + ya;
    // e.g. 5x+2 for 1280 -> 256) the kernels compute them instead of loading them
    int affine, xm, xa, ym, ya;
};

// Resized value (0..255) of output pixel (y, x), memory channel cc, of one source frame.
__device__ __forceinline__ int sample_u8(const uint8_t* __restrict__ frame, int y, int x, int cc,
                                         const SampleArgs& a) {
    if (a.mode == 1) {
        return frame[(int64_t)a.sy[y] * a.row_stride + a.sx[x] * 3 + cc];
    }
    if (a.mode == 2) {
        const uint8_t* r0 = frame + (int64_t)(2 * y) * a.row_stride + (2 * x) * 3 + cc;
        const uint8_t* r1 = r0 + a.row_stride;
        return (r0[0] + r0[3] + r1[0] + r1[3] + 2) >> 2;
    }
    int sy = a.sy[y];
    int r0 = min(max(sy, 0), a.H0 - 1), r1 = min(max(sy + 1, 0), a.H0 - 1);
    int c0 = a.sx[x], c1 = min(c0 + 1, a.W0 - 1);
    int a0 = a.ax[2 * x], a1 = a.ax[2 * x + 1];
    int b0 = a.by[2 * y], b1 = a.by[2 * y + 1];
    const uint8_t* p0 = frame + (int64_t)r0 * a.row_stride + cc;
    const uint8_t* p1 = frame + (int64_t)r1 * a.row_stride + cc;
    int S0 = p0[c0 * 3] * a0 + p0[c1 * 3] * a1;
    int S1 = p1[c0 * 3] * a0 + p1[c1 * 3] * a1;
    int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    return min(max(v, 0), 255);
}

}  // namespace cd

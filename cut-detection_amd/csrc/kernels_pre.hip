// Stand-alone resize+normalise kernel and the synthetic clip generator.
#include <algorithm>

#include "cd_sample.h"

namespace cd {

// ------------------------------------------------------------------------------------------
// Resize + normalise: one workgroup per (frame, output row).
//
// HBM-bound: the only compulsory traffic is the source rows that are actually sampled
// (read once, 16 B per lane, into LDS) and the float32 planes written.  The bilinear taps are
// then picked out of LDS, so the byte-granular, stride-15/22 access pattern never reaches the
// vector memory pipe.  Output is what reference frameID/data.py:225-228 produces: float32 CHW,
// RGB order, value/255 (through a table of correctly rounded k/255).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_rows_kernel(
    const uint8_t* __restrict__ src, SampleArgs a, int64_t frame_stride, int chan_order,
    float* __restrict__ dst, int nh, int nw, const float* __restrict__ lut, int row_bytes_padded) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rows[];   // [2][row_bytes_padded]
    __shared__ float slut[256];
    const int tid = threadIdx.x;
    const int y = blockIdx.x % nh;
    const int f = blockIdx.x / nh;
    const uint8_t* frame = src + (int64_t)f * frame_stride;
    slut[tid] = lut[tid];

    int r0, r1;
    if (a.mode == 2) { r0 = 2 * y; r1 = 2 * y + 1; }
    else {
        int sy = a.sy[y];
        r0 = min(max(sy, 0), a.H0 - 1);
        r1 = min(max(sy + 1, 0), a.H0 - 1);
    }
    const int b0 = a.mode == 0 ? a.by[2 * y] : 0, b1 = a.mode == 0 ? a.by[2 * y + 1] : 0;
    const bool two_rows = a.mode == 2 || (a.mode == 0 && b1 != 0);
    const int row_bytes = a.W0 * 3;
    const int chunks = row_bytes >> 4;
    {
        const uint4* g0 = reinterpret_cast<const uint4*>(frame + (int64_t)r0 * a.row_stride);
        const uint4* g1 = reinterpret_cast<const uint4*>(frame + (int64_t)r1 * a.row_stride);
        uint4* l0 = reinterpret_cast<uint4*>(rows);
        uint4* l1 = reinterpret_cast<uint4*>(rows + row_bytes_padded);
        for (int c = tid; c < chunks; c += 256) {
            l0[c] = g0[c];
            if (two_rows) l1[c] = g1[c];
        }
        for (int b = (chunks << 4) + tid; b < row_bytes; b += 256) {   // < 16 tail bytes
            rows[b] = frame[(int64_t)r0 * a.row_stride + b];
            if (two_rows) rows[row_bytes_padded + b] = frame[(int64_t)r1 * a.row_stride + b];
        }
    }
    __syncthreads();
    const uint8_t* R0 = rows;
    const uint8_t* R1 = two_rows ? rows + row_bytes_padded : rows;
    const int64_t plane = (int64_t)nh * nw;
    float* out = dst + (int64_t)f * 3 * plane + (int64_t)y * nw;
    for (int x = tid; x < nw; x += 256) {
        int v[3];
        if (a.mode == 1) {
            int c0 = a.sx[x] * 3;
            for (int cc = 0; cc < 3; ++cc) v[cc] = R0[c0 + cc];
        } else if (a.mode == 2) {
            int c0 = 6 * x;
            for (int cc = 0; cc < 3; ++cc)
                v[cc] = (R0[c0 + cc] + R0[c0 + 3 + cc] + R1[c0 + cc] + R1[c0 + 3 + cc] + 2) >> 2;
        } else {
            int c0 = a.sx[x], c1 = min(c0 + 1, a.W0 - 1);
            int a0 = a.ax[2 * x], a1 = a.ax[2 * x + 1];
            for (int cc = 0; cc < 3; ++cc) {
                int S0 = R0[c0 * 3 + cc] * a0 + R0[c1 * 3 + cc] * a1;
                int S1 = R1[c0 * 3 + cc] * a0 + R1[c1 * 3 + cc] * a1;
                int t = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
                v[cc] = min(max(t, 0), 255);
            }
        }
        // plane 0 is R: memory channel 2 when the frame is BGR
        for (int ci = 0; ci < 3; ++ci) out[ci * plane + x] = slut[v[chan_order == CD_BGR ? 2 - ci : ci]];
    }
}

// Fallback for rows that are not 16-byte aligned: taps straight from global memory.
__global__ __launch_bounds__(256) void preprocess_generic_kernel(
    const uint8_t* __restrict__ src, SampleArgs a, int64_t frame_stride, int chan_order,
    float* __restrict__ dst, int nh, int nw, const float* __restrict__ lut) {
    const int y = blockIdx.x % nh;
    const int f = blockIdx.x / nh;
    const uint8_t* frame = src + (int64_t)f * frame_stride;
    const int64_t plane = (int64_t)nh * nw;
    float* out = dst + (int64_t)f * 3 * plane + (int64_t)y * nw;
    for (int x = threadIdx.x; x < nw; x += 256)
        for (int ci = 0; ci < 3; ++ci)
            out[ci * plane + x] = lut[sample_u8(frame, y, x, chan_order == CD_BGR ? 2 - ci : ci, a)];
}

int launch_preprocess(const cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                      int64_t frame_stride, int chan_order, float* dst, int nh, int nw, hipStream_t s) {
    SampleArgs a{h->rt_pre.sx, h->rt_pre.ax, h->rt_pre.sy, h->rt_pre.by, h->rt_pre.mode, H0, W0, row_stride, 0, 0, 0, 0, 0};
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 15) == 0;
    dim3 grid((unsigned)(n * nh)), block(256);
    if (aligned) {
        int padded = (W0 * 3 + 15) & ~15;
        hipLaunchKernelGGL(preprocess_rows_kernel, grid, block, (size_t)2 * padded, s, src, a, frame_stride,
                           chan_order, dst, nh, nw, h->p.lut, padded);
    } else {
        hipLaunchKernelGGL(preprocess_generic_kernel, grid, block, 0, s, src, a, frame_stride, chan_order, dst,
                           nh, nw, h->p.lut);
    }
    CD_HIP(hipGetLastError());
    return CD_OK;
}

// ------------------------------------------------------------------------------------------
// Resize only, u8 in -> u8 out (cv2.resize at reference frameID/data.py:220-222 by itself): the same
// row staging as preprocess_rows_kernel, the resized pixels written interleaved in the source's
// channel order.  One workgroup per (frame, output row): at 1080p 2 x 5760 bytes in, 768 out.
// ------------------------------------------------------------------------------------------
// CELLS: the resized pixels are written as the first block's matrix-core operand cells instead of bytes
// - 8 bytes per pixel, {R, G, B, 0} as bfloat16 (a byte value is exact in bfloat16) - into frames with a
// zero margin: [kCellPad + nh + kCellPad rows][kCellPitch cells], pixel (y, x) at row y + kCellPad, cell
// x + kCellPad; the margin is zeroed when the buffer is allocated and never written.
// Workgroups of kResizeThreads threads.  Alone, the kernel is bound by the source bytes a CU has in flight,
// i.e. by its resident workgroups (each holds one output row's two source rows): with 256 threads the
// 2048-thread limit of a CU caps them at 8 (1080p, 512 frames: 206 us, 4.1 TB/s), with 128 the LDS caps
// them at 13 (186 us).  In the range pipeline it runs beside the convolution kernels, and there the 13
// workgroups' 150 KB of LDS per CU cost more than the faster resize gains (1080p: bf16 1.29 -> 1.22 M
// frames/s, split bf16 860 -> 851 k): 256 it is.
#ifndef CD_RESIZE_THREADS
#define CD_RESIZE_THREADS 256
#endif
constexpr int kResizeThreads = CD_RESIZE_THREADS;
template <bool CELLS>
__global__ __launch_bounds__(kResizeThreads) void resize_u8_rows_kernel(
    const uint8_t* __restrict__ src, SampleArgs a, int64_t frame_stride, uint8_t* __restrict__ dst, int nh, int nw,
    int row_bytes_padded, int chan_order) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rows[];   // [2][row_bytes_padded]
    const int tid = threadIdx.x;
    const int y = blockIdx.x % nh;
    const int f = blockIdx.x / nh;
    const uint8_t* frame = src + (int64_t)f * frame_stride;
    int r0, r1;
    if (a.mode == 2) { r0 = 2 * y; r1 = 2 * y + 1; }
    else {
        int sy = a.sy[y];
        r0 = min(max(sy, 0), a.H0 - 1);
        r1 = min(max(sy + 1, 0), a.H0 - 1);
    }
    const int b0 = a.mode == 0 ? a.by[2 * y] : 0, b1 = a.mode == 0 ? a.by[2 * y + 1] : 0;
    const bool two_rows = a.mode == 2 || (a.mode == 0 && b1 != 0);
    const int row_bytes = a.W0 * 3;
    const int chunks = row_bytes >> 4;
    {
        const uint4* g0 = reinterpret_cast<const uint4*>(frame + (int64_t)r0 * a.row_stride);
        const uint4* g1 = reinterpret_cast<const uint4*>(frame + (int64_t)r1 * a.row_stride);
        uint4* l0 = reinterpret_cast<uint4*>(rows);
        uint4* l1 = reinterpret_cast<uint4*>(rows + row_bytes_padded);
        for (int c = tid; c < chunks; c += kResizeThreads) {
            l0[c] = g0[c];
            if (two_rows) l1[c] = g1[c];
        }
        for (int b = (chunks << 4) + tid; b < row_bytes; b += kResizeThreads) {   // < 16 tail bytes
            rows[b] = frame[(int64_t)r0 * a.row_stride + b];
            if (two_rows) rows[row_bytes_padded + b] = frame[(int64_t)r1 * a.row_stride + b];
        }
    }
    __syncthreads();
    const uint8_t* R0 = rows;
    const uint8_t* R1 = two_rows ? rows + row_bytes_padded : rows;
    uint8_t* out = dst + ((int64_t)f * nh + y) * nw * 3;
    for (int x = tid; x < nw; x += kResizeThreads) {
        int v[3];
        if (a.mode == 1) {
            int c0 = a.sx[x] * 3;
            for (int cc = 0; cc < 3; ++cc) v[cc] = R0[c0 + cc];
        } else if (a.mode == 2) {
            int c0 = 6 * x;
            for (int cc = 0; cc < 3; ++cc)
                v[cc] = (R0[c0 + cc] + R0[c0 + 3 + cc] + R1[c0 + cc] + R1[c0 + 3 + cc] + 2) >> 2;
        } else {
            int c0 = a.sx[x], c1 = min(c0 + 1, a.W0 - 1);
            int a0 = a.ax[2 * x], a1 = a.ax[2 * x + 1];
            for (int cc = 0; cc < 3; ++cc) {
                int S0 = R0[c0 * 3 + cc] * a0 + R0[c1 * 3 + cc] * a1;
                int S1 = R1[c0 * 3 + cc] * a0 + R1[c1 * 3 + cc] * a1;
                int t = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
                v[cc] = min(max(t, 0), 255);
            }
        }
        if constexpr (CELLS) {
            const bool bgr = chan_order == CD_BGR;
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t rg = {(__bf16)(float)v[bgr ? 2 : 0], (__bf16)(float)v[1]};
            const bf16x2_t b0 = {(__bf16)(float)v[bgr ? 0 : 2], (__bf16)0.f};
            uint2* cells = reinterpret_cast<uint2*>(dst) +
                           ((int64_t)f * (nh + 2 * kCellPad) + y + kCellPad) * kCellPitch + kCellPad;
            cells[x] = uint2{__builtin_bit_cast(uint32_t, rg), __builtin_bit_cast(uint32_t, b0)};
        } else {
            for (int cc = 0; cc < 3; ++cc) out[3 * x + cc] = (uint8_t)v[cc];
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same resize without LDS: one thread per output pixel, its taps read straight from global memory
// as unaligned dwords (the six bytes of two neighbouring source pixels sit in eight), kDirectRows output
// rows per thread so that a wave has 4 x kDirectRows loads in flight.  Neighbouring lanes' taps are a few
// bytes apart, so a wave-wide load touches about as many cache lines as a row copy would, and every
// source line is still fetched once.  Why it exists: the row-staging kernel above holds 11.5 KB of LDS
// per workgroup, and in the range pipeline the convolution kernels leave a CU no LDS to give - its
// workgroups then wait for them instead of running beside them.  This one needs registers only.
// The three output bytes of four neighbouring pixels leave as three dwords (lane exchange, no LDS memory).
// ------------------------------------------------------------------------------------------
constexpr int kDirectRows = 4;

// the 8 bytes at row + off (any alignment), shifted so that byte 0 is row[off]; reads stay inside the row:
// near its end the window slides back and the value is shifted down instead (row_bytes >= 8)
__device__ __forceinline__ uint64_t load8_in_row(const uint8_t* row, int off, int row_bytes) {
    typedef uint32_t __attribute__((aligned(1))) u32_any;
    const int at = min(off, row_bytes - 8);
    const uint32_t lo = *reinterpret_cast<const u32_any*>(row + at), hi = *reinterpret_cast<const u32_any*>(row + at + 4);
    return ((uint64_t)hi << 32 | lo) >> (8 * (off - at));
}

// Grid: x = 64-pixel column blocks, y = blocks of 4 row groups (one per wave), z = frame.
template <int MODE, bool CELLS>
__global__ __launch_bounds__(256) void resize_u8_direct_kernel(
    const uint8_t* __restrict__ src, SampleArgs a, int64_t frame_stride, uint8_t* __restrict__ dst, int nh, int nw,
    int chan_order) {
    const int x = min((int)(blockIdx.x * 64 + threadIdx.x), nw - 1);        // surplus lanes repeat the last pixel, unstored
    const bool x_live = (int)(blockIdx.x * 64 + threadIdx.x) < nw;
    // (a wave is one row group: everything that depends on the row only is made scalar work)
    const int y0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + threadIdx.y) * kDirectRows);
    const int f = blockIdx.z;
    if (y0 >= nh) return;                                                   // whole waves: a wave is one row group
    const uint8_t* frame = src + (int64_t)f * frame_stride;
    const int row_bytes = a.W0 * 3;
    int off;
    uint32_t coef = 0;                           // bilinear: the horizontal coefficient pair, a0 | a1 << 16
    // byte selectors that put source bytes cc and cc + 3 (the pixel and its right neighbour) into the two
    // halves of a dword; at the right edge the neighbour is the pixel itself
    uint32_t sel[3] = {0x0c030c00u, 0x0c040c01u, 0x0c050c02u};
    if (MODE == 1) off = a.sx[x] * 3;
    else if (MODE == 2) off = 6 * x;
    else {
        const int c0 = a.sx[x];
        off = c0 * 3;
        coef = *reinterpret_cast<const uint32_t*>(a.ax + 2 * x);
        if (c0 + 1 > a.W0 - 1) { sel[0] = 0x0c000c00u; sel[1] = 0x0c010c01u; sel[2] = 0x0c020c02u; }
    }
    uint64_t t0[kDirectRows], t1[kDirectRows];
    int b0[kDirectRows], b1[kDirectRows];
#pragma unroll
    for (int r = 0; r < kDirectRows; ++r) {
        const int y = min(y0 + r, nh - 1);       // rows past the end repeat the last one and are not stored
        int r0, r1;
        if (MODE == 2) { r0 = 2 * y; r1 = 2 * y + 1; b0[r] = b1[r] = 0; }
        else {
            const int sy = a.sy[y];
            r0 = min(max(sy, 0), a.H0 - 1);
            r1 = min(max(sy + 1, 0), a.H0 - 1);
            b0[r] = MODE == 0 ? a.by[2 * y] : 0;
            b1[r] = MODE == 0 ? a.by[2 * y + 1] : 0;
        }
        t0[r] = load8_in_row(frame + (int64_t)r0 * a.row_stride, off, row_bytes);
        // (a zero vertical weight makes the second row's value irrelevant; reading it anyway keeps the loads
        // unconditional, and r1 is a valid row)
        t1[r] = MODE == 1 ? t0[r] : load8_in_row(frame + (int64_t)r1 * a.row_stride, off, row_bytes);
    }
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int r = 0; r < kDirectRows; ++r) {
        const int y = y0 + r;
        int v[3];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            if (MODE == 1) v[cc] = (int)(t0[r] >> (8 * cc)) & 255;
            else {
                const uint32_t q0 = __builtin_amdgcn_perm((uint32_t)(t0[r] >> 32), (uint32_t)t0[r], sel[cc]);
                const uint32_t q1 = __builtin_amdgcn_perm((uint32_t)(t1[r] >> 32), (uint32_t)t1[r], sel[cc]);
                if (MODE == 2) {
                    const uint32_t sum = q0 + q1;                          // two 16-bit lanes, no carry between them
                    v[cc] = (int)((sum & 0xFFFFu) + (sum >> 16) + 2) >> 2;
                } else {
                    const int S0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, q0), __builtin_bit_cast(u16x2, coef), 0u, false);
                    const int S1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, q1), __builtin_bit_cast(u16x2, coef), 0u, false);
                    // (24-bit multiplies: |b| <= 2^11 and S >> 4 < 2^15 - the full 32-bit multiply runs at a quarter of the rate)
                    const int t = ((__mul24(b0[r], S0 >> 4) >> 16) + (__mul24(b1[r], S1 >> 4) >> 16) + 2) >> 2;
                    v[cc] = min(max(t, 0), 255);
                }
            }
        }
        if constexpr (CELLS) {
            const bool bgr = chan_order == CD_BGR;
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t rg2 = {(__bf16)(float)v[bgr ? 2 : 0], (__bf16)(float)v[1]};
            const bf16x2_t b2 = {(__bf16)(float)v[bgr ? 0 : 2], (__bf16)0.f};
            uint2* cells = reinterpret_cast<uint2*>(dst) +
                           ((int64_t)f * (nh + 2 * kCellPad) + y + kCellPad) * kCellPitch + kCellPad;
            if (x_live && y < nh) cells[x] = uint2{__builtin_bit_cast(uint32_t, rg2), __builtin_bit_cast(uint32_t, b2)};
        } else {
            // pixels 4q .. 4q+3 of a row are lanes 4q .. 4q+3 of the wave (nw is a multiple of 4): lanes 0, 1, 2
            // of the quad store dwords 0, 1, 2 of its 12 bytes
            const uint32_t own = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16;
            const uint32_t nxt = (uint32_t)__shfl_down((int)own, 1, 64);
            const int q = threadIdx.x & 3;
            const uint32_t d = q == 0 ? own | nxt << 24 : q == 1 ? own >> 8 | nxt << 16 : own >> 16 | nxt << 8;
            uint32_t* out = reinterpret_cast<uint32_t*>(dst + ((int64_t)f * nh + y) * nw * 3) + (x >> 2) * 3 + q;
            if (x_live && y < nh && q < 3) *out = d;
        }
    }
}

// launches the direct kernel for the table's mode; false when it does not apply (then the callers fall back)
template <bool CELLS>
static bool launch_resize_direct(const SampleArgs& a, const uint8_t* src, int64_t frame_stride, uint8_t* dst, int n,
                                 int nh, int nw, int chan_order, hipStream_t s) {
    if (nw % 4 != 0 || a.W0 * 3 < 8 || (reinterpret_cast<uintptr_t>(dst) & 3) != 0 || n > 65535) return false;
    if (a.mode == 0 && (reinterpret_cast<uintptr_t>(a.ax) & 3) != 0) return false;
    const int row_groups = (nh + kDirectRows - 1) / kDirectRows;
    const dim3 grid((unsigned)((nw + 63) / 64), (unsigned)((row_groups + 3) / 4), (unsigned)n), block(64, 4);
    if (a.mode == 0)
        hipLaunchKernelGGL((resize_u8_direct_kernel<0, CELLS>), grid, block, 0, s, src, a, frame_stride, dst, nh, nw, chan_order);
    else if (a.mode == 1)
        hipLaunchKernelGGL((resize_u8_direct_kernel<1, CELLS>), grid, block, 0, s, src, a, frame_stride, dst, nh, nw, chan_order);
    else
        hipLaunchKernelGGL((resize_u8_direct_kernel<2, CELLS>), grid, block, 0, s, src, a, frame_stride, dst, nh, nw, chan_order);
    return true;
}

// rows that are not 16-byte aligned: taps straight from global memory
__global__ __launch_bounds__(256) void resize_u8_generic_kernel(
    const uint8_t* __restrict__ src, SampleArgs a, int64_t frame_stride, uint8_t* __restrict__ dst, int nh, int nw) {
    const int y = blockIdx.x % nh;
    const int f = blockIdx.x / nh;
    const uint8_t* frame = src + (int64_t)f * frame_stride;
    uint8_t* out = dst + ((int64_t)f * nh + y) * nw * 3;
    for (int x = threadIdx.x; x < nw; x += 256)
        for (int cc = 0; cc < 3; ++cc) out[3 * x + cc] = (uint8_t)sample_u8(frame, y, x, cc, a);
}

int launch_resize_u8(const cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                     int64_t frame_stride, uint8_t* dst, int nh, int nw, hipStream_t s) {
    SampleArgs a{h->rt_pre.sx, h->rt_pre.ax, h->rt_pre.sy, h->rt_pre.by, h->rt_pre.mode, H0, W0, row_stride, 0, 0, 0, 0, 0};
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 15) == 0;
    const int padded = (W0 * 3 + 15) & ~15;
    dim3 grid((unsigned)(n * nh)), block(256);
    if (!h->opt[CD_OPT_RESIZE_ROWS] && launch_resize_direct<false>(a, src, frame_stride, dst, n, nh, nw, 0, s)) {
    } else if (aligned && 2 * padded <= 64 * 1024)
        hipLaunchKernelGGL(resize_u8_rows_kernel<false>, grid, dim3(kResizeThreads), (size_t)2 * padded, s, src, a,
                           frame_stride, dst, nh, nw, padded, 0);
    else
        hipLaunchKernelGGL(resize_u8_generic_kernel, grid, block, 0, s, src, a, frame_stride, dst, nh, nw);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

// The resize with operand cells as output (see resize_u8_rows_kernel<true>); `cells` points at the first
// frame of the batch inside a buffer laid out by cell_frame_bytes().  Needs 16-byte aligned source rows
// that fit the kernel's LDS (the caller checks with resize_cells_supported).
bool resize_cells_supported(const uint8_t* src, int W0, int64_t row_stride, int64_t frame_stride, int nw) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 15) == 0;
    return aligned && 2 * ((W0 * 3 + 15) & ~15) <= 64 * 1024 && nw == kNetW;
}

int launch_resize_cells(const cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                        int64_t frame_stride, int chan_order, uint8_t* cells, int nh, int nw, hipStream_t s) {
    SampleArgs a{h->rt_pre.sx, h->rt_pre.ax, h->rt_pre.sy, h->rt_pre.by, h->rt_pre.mode, H0, W0, row_stride, 0, 0, 0, 0, 0};
    const int padded = (W0 * 3 + 15) & ~15;
    if (!h->opt[CD_OPT_RESIZE_ROWS] && launch_resize_direct<true>(a, src, frame_stride, cells, n, nh, nw, chan_order, s)) {
    } else
    hipLaunchKernelGGL(resize_u8_rows_kernel<true>, dim3((unsigned)(n * nh)), dim3(kResizeThreads), (size_t)2 * padded, s, src, a,
                       frame_stride, cells, nh, nw, padded, chan_order);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

// ------------------------------------------------------------------------------------------
// NV12 -> interleaved BGR/RGB u8, optionally only selected rows (HBM-bound: 1.5 bytes in, 3 out per
// pixel).  Arithmetic: ITU-R BT.601 limited range with 20-bit coefficients, as OpenCV's
// cvtColor(COLOR_YUV2BGR_NV12) publishes it - the conversion cv2.VideoCapture.read() has already
// applied to the frames the reference sees (frameID/data.py:213).
// One thread converts 4 pixels: one dword of luma, one dword of chroma (two U,V pairs), three
// dwords out, so a wave reads 256 + 256 contiguous bytes and writes 768.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sat8(int v) { return (uint32_t)min(max(v, 0), 255); }

__device__ __forceinline__ uint32_t nv12_pixel(int Y, int U, int V, bool bgr) {
    const int y = max(0, Y - 16) * 1220542 + (1 << 19);
    const uint32_t r = sat8((y + 1673527 * V) >> 20);
    const uint32_t g = sat8((y - 852492 * V - 409993 * U) >> 20);
    const uint32_t b = sat8((y + 2116026 * U) >> 20);
    return bgr ? (b | (g << 8) | (r << 16)) : (r | (g << 8) | (b << 16));
}

__global__ __launch_bounds__(256) void nv12_to_bgr_kernel(
    const uint8_t* __restrict__ yp, const uint8_t* __restrict__ uvp, int W0, int64_t y_rs, int64_t uv_rs,
    int64_t y_fs, int64_t uv_fs, const int32_t* __restrict__ y_rows, const int32_t* __restrict__ uv_rows,
    int n_rows, uint8_t* __restrict__ dst, int chan_order, int quads, bool aligned) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y, f = blockIdx.z;
    if (q >= quads) return;
    const int yr = y_rows ? y_rows[r] : r, ur = uv_rows ? uv_rows[r] : (yr >> 1);
    const uint8_t* ys = yp + (int64_t)f * y_fs + (int64_t)yr * y_rs + 4 * q;
    const uint8_t* us = uvp + (int64_t)f * uv_fs + (int64_t)ur * uv_rs + 4 * q;
    uint8_t* out = dst + (((int64_t)f * n_rows + r) * W0 + 4 * q) * 3;
    const bool bgr = chan_order == CD_BGR;
    const int left = W0 - 4 * q;                 // pixels this thread owns (4, or fewer at the row end)
    uint32_t y4 = 0, uv4 = 0;
    if (aligned && left >= 4) {
        y4 = *reinterpret_cast<const uint32_t*>(ys);
        uv4 = *reinterpret_cast<const uint32_t*>(us);
    } else {
        for (int k = 0; k < min(left, 4); ++k) y4 |= (uint32_t)ys[k] << (8 * k);
        for (int k = 0; k < min((left + 1) & ~1, 4); ++k) uv4 |= (uint32_t)us[k] << (8 * k);
    }
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int U = (int)((uv4 >> (16 * (k >> 1))) & 255) - 128, V = (int)((uv4 >> (16 * (k >> 1) + 8)) & 255) - 128;
        px[k] = nv12_pixel((int)((y4 >> (8 * k)) & 255), U, V, bgr);
    }
    if (aligned && left >= 4) {
        uint32_t* o = reinterpret_cast<uint32_t*>(out);
        o[0] = px[0] | (px[1] << 24);
        o[1] = (px[1] >> 8) | (px[2] << 16);
        o[2] = (px[2] >> 16) | (px[3] << 8);
    } else {
        for (int k = 0; k < min(left, 4); ++k)
            for (int c = 0; c < 3; ++c) out[3 * k + c] = (uint8_t)(px[k] >> (8 * c));
    }
}

int launch_nv12_to_bgr(const uint8_t* y, const uint8_t* uv, int n, int W0, int64_t y_rs, int64_t uv_rs,
                       int64_t y_fs, int64_t uv_fs, const int32_t* y_rows, const int32_t* uv_rows, int n_rows,
                       uint8_t* dst, int chan_order, hipStream_t s) {
    const int quads = (W0 + 3) / 4;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(uv) |
                           reinterpret_cast<uintptr_t>(dst) | (uintptr_t)y_rs | (uintptr_t)uv_rs | (uintptr_t)y_fs |
                           (uintptr_t)uv_fs | (uintptr_t)W0;
    dim3 grid((unsigned)((quads + 255) / 256), (unsigned)n_rows, (unsigned)n), block(256);
    hipLaunchKernelGGL(nv12_to_bgr_kernel, grid, block, 0, s, y, uv, W0, y_rs, uv_rs, y_fs, uv_fs, y_rows, uv_rows,
                       n_rows, dst, chan_order, quads, (bits & 3) == 0);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

// ------------------------------------------------------------------------------------------
// Synthetic clip generator: integer hash, four bytes per thread.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ dst, int H0, int W0,
                                                    const uint8_t* __restrict__ classes, uint32_t seed,
                                                    int64_t first_index, uint32_t frame_elems,
                                                    int period, int thick) {
    const int f = blockIdx.y;
    const uint32_t e0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (e0 >= frame_elems) return;
    const int cls = classes[f];
    const uint32_t key = mix32(seed + 0x9E3779B9u * (uint32_t)((first_index + f + 1) & 0xFFFFFFFFll));
    uint8_t* out = dst + (int64_t)f * frame_elems + e0;
    uint32_t packed = 0;
    const uint32_t row_elems = (uint32_t)W0 * 3u;
    for (uint32_t k = 0; k < 4; ++k) {
        uint32_t e = e0 + k;
        if (e >= frame_elems) break;
        uint32_t r = mix32(key ^ (e * 0x85EBCA6Bu + 0xC2B2AE35u));
        uint32_t v;
        if (cls == 0) v = r >> 24;
        else if (cls == 2) v = r >> 29;
        else {
            uint32_t y = e / row_elems, c = e % 3u;
            uint32_t base = (y % (uint32_t)period) < (uint32_t)thick ? 230u : (c == 1u ? 128u : 26u);
            v = base + (r >> 29);
        }
        packed |= v << (8 * k);
    }
    if (e0 + 4 <= frame_elems) *reinterpret_cast<uint32_t*>(out) = packed;
    else for (uint32_t k = 0; e0 + k < frame_elems; ++k) out[k] = (uint8_t)(packed >> (8 * k));
}

int launch_synth(uint8_t* dst, int n, int H0, int W0, const uint8_t* classes, uint32_t seed,
                 int64_t first_index, hipStream_t s) {
    const uint64_t elems = (uint64_t)H0 * W0 * 3;
    if (elems >= (1ull << 32) || (elems & 3)) {
        set_error("cd_synth_frames: frame of %llu bytes unsupported (must be < 4 GiB and a multiple of 4)",
                  (unsigned long long)elems);
        return CD_EINVAL;
    }
    dim3 grid((unsigned)((elems / 4 + 255) / 256), (unsigned)n), block(256);
    // stripe period and thickness in source rows (12 and 1 network rows); narrow frames keep a period of 1
    hipLaunchKernelGGL(synth_kernel, grid, block, 0, s, dst, H0, W0, classes, seed, first_index,
                       (uint32_t)elems, std::max(1, (12 * W0) / 256), (W0 + 255) / 256);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

}  // namespace cd

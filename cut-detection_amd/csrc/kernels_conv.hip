// prod_net forward: three fused conv3x3 -> ReLU -> maxpool3 -> BatchNorm blocks and the head.
//
// Every convolution is an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 fma chain):
//     M = output pixels, N = output channels (48 = 3 tiles of 16), K = (ci, ky, kx).
// Operand lane maps:  A: lane l holds A[row l&15][k l>>4];  B: lane l holds B[k l>>4][col l&15];
//                     D: lane l holds D[row 4*(l>>4)+r][col l&15] in register r (r = 0..3).
//
// The unit of work is a WAVE TILE: 16 max-pool windows x 48 channels, i.e. 144 conv pixels =
// 9 M-tiles (t = conv row inside the pool row, j = group of 4 columns) x 3 N-tiles = 27
// accumulators.  M-row m = 4q + r of tile (t, j) is mapped to the conv pixel in pool-row slot q,
// conv row t, column 4j + r of that slot's 12 columns, so that the four registers of lane
// (q, channel) across the nine tiles hold exactly the 3x3 windows of 4 pool outputs: ReLU, the
// 3x3 max, the bias and the BatchNorm affine all happen in registers and only the pooled
// tensor is ever written (the reference materialises the full conv output, net.py:35-38).
//
// Activations between blocks are channel-last fp32 with a one-pixel zero border stored in HBM
// ([frame][rows+2][cols+2][48]; the border is zeroed when the workspace is allocated and never
// written), so that the next block's 3x3 taps need no bounds checks, its input tile is a plain
// strided copy (LDS-DMA), and one 16-byte LDS read gives a lane the A values of four
// consecutive k-steps.
//
// Workgroup -> work mapping is XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so
// workgroup b works on frame (b % 8) + 8 * (...): all tiles of one frame run on one XCD and
// share its L2 for tile halos and weights.  This only affects speed.
#include "cd_sample.h"

namespace cd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// Storage type of the activations between blocks and of the block-2/3 weights.  A fragment is a
// lane's share of an MFMA operand for one 16-wide K step: four fp32 values feed four
// v_mfma_f32_16x16x4_f32 (K slots 4e + kq), four bf16 values feed one v_mfma_f32_16x16x16_bf16
// (K slots 4kq + e) - the same four input channels either way, so both precisions share every
// index computation.  Accumulation is fp32 in both.
//
// A third arithmetic, split bf16, keeps fp32's storage size and every address: each fp32 value v
// is stored as two bfloat16, hi = bf16(v) and lo = bf16(v - hi), and the 16-byte cell that holds
// four fp32 channels in fp32 mode holds {hi x 4, lo x 4} of the same four channels.  That cell is,
// as it stands, a lane's A operand of v_mfma_f32_16x16x32_bf16 (8 K slots per lane: which channel
// sits in which slot is free as long as A and B agree).  With B operands {w_hi x 4,
// w_hi x 4} and {w_lo x 4, w_lo x 4}, two such MFMAs give
//     (a_hi + a_lo) * w_hi + (a_hi + a_lo) * w_lo,
// the complete product of the split values, in 2 x 16 cycles where fp32 needs 4 x 32.  What is lost
// is only the splitting itself: |v - (hi + lo)| <= 2^-17 |v|.
constexpr int kF32 = 0, kBF16 = 1, kS2 = 2;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifdef CD_S2_WDUP                     // measurement build: B operands stored ready-made (twice the bytes, two loads, no moves)
struct SplitB { u32x4 h, l; };
#else
struct SplitB { u32x4 v; };          // weights of one (step, n-tile) as loaded: {w_hi x 4, w_lo x 4} per lane
#endif
template <int P> struct Prec;
template <> struct Prec<kF32> {
    typedef float elem; typedef f32x4 frag; typedef f32x4 bfrag;
    static constexpr int kSub = 4;          // MFMAs per fragment pair
    static constexpr int kEs = 4;           // bytes per element
    static constexpr int kWBytes = 16;      // bytes of one lane's B fragment
#ifndef CD_F32_BDIST
#define CD_F32_BDIST 1
#endif
    static constexpr int kBDist = CD_F32_BDIST;   // block 2: steps a weight fragment is requested ahead of its use
    static constexpr int kBLoads = 3;       // block 2: buffer loads per step
};
template <> struct Prec<kBF16> {
    typedef uint16_t elem; typedef bf16x4 frag; typedef bf16x4 bfrag;
    static constexpr int kSub = 1;
    static constexpr int kEs = 2;
    static constexpr int kWBytes = 8;
    static constexpr int kBDist = 1;
    static constexpr int kBLoads = 3;
};
template <> struct Prec<kS2> {
    typedef uint32_t elem; typedef u32x4 frag; typedef SplitB bfrag;   // element = one channel's (hi, lo), by size only
#ifdef CD_EXP_S2_SUB       // timing experiment: more MFMAs per fragment pair (what three-term activations would issue)
    static constexpr int kSub = CD_EXP_S2_SUB;
#else
    static constexpr int kSub = 2;
#endif
    static constexpr int kEs = 4;
#ifdef CD_S2_WDUP
    static constexpr int kWBytes = 32;
#else
    static constexpr int kWBytes = 16;
#endif
#ifndef CD_S2_BDIST
#define CD_S2_BDIST 2
#endif
    static constexpr int kBDist = CD_S2_BDIST;   // a step is only ~860 cycles here: one step ahead does not cover an L2 round trip
#ifdef CD_S2_WDUP
    static constexpr int kBLoads = 6;
#else
    static constexpr int kBLoads = 3;
#endif
};

// Weight fragments are read through a buffer descriptor: the address is one per-lane VGPR
// (lane * fragment bytes) plus a scalar offset, so the 81 loads of a tile cost no vector address
// arithmetic and nothing for the compiler to hoist into registers.
template <int P>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const void* w) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, 27 * 3 * 64 * Prec<P>::kWBytes, 0x00020000);
}
// fragment `idx` (= step * 3 + n-tile) of a layer's packed weights
template <int P>
__device__ __forceinline__ typename Prec<P>::bfrag load_frag(__amdgpu_buffer_rsrc_t r, int lane, int idx) {
    if constexpr (P == kBF16)
        return __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, idx * 512, 0));
    else if constexpr (P == kS2) {    // [idx][lane]{hi x 4, lo x 4}: one load; mfma_frag builds the operands {hi, hi} / {lo, lo}
#ifdef CD_EXP_WHOT
        idx &= 1;                     // timing experiment: every weight load hits L1 (wrong results)
#endif
#ifdef CD_S2_WDUP
        return SplitB{__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, idx * 2048, 0)),
                      __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, idx * 2048 + 1024, 0))};
#else
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, idx * 1024, 0));
        return SplitB{v};
#endif
    } else
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, idx * 1024, 0));
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// sub-step e of a fragment pair (fp32: e = 0..3; bf16: e = 0 only; split bf16: e = 0 the w_lo
// terms, e = 1 the w_hi terms)
template <int P>
__device__ __forceinline__ f32x4 mfma_frag(const typename Prec<P>::frag& a, const typename Prec<P>::bfrag& b, int e,
                                           f32x4 c) {
    if constexpr (P == kBF16) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
#ifdef CD_S2_WDUP
    else if constexpr (P == kS2)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, e == 0 ? b.l : b.h), c, 0, 0, 0);
#else
    else if constexpr (P == kS2)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, e == 0 ? u32x4{b.v[2], b.v[3], b.v[2], b.v[3]}
                                                                                         : u32x4{b.v[0], b.v[1], b.v[0], b.v[1]}),
                                                       c, 0, 0, 0);
#endif
    else return mfma4(a[e], b[e], c);
}
// two fp32 values -> two bf16 (round to nearest even) in one dword, `lo` in the low half
// (compiles to v_cvt_pk_bf16_f32; written as a conversion, not inline asm, so that the compiler
// keeps its own hazard bookkeeping around it)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// four fp32 values -> one split-bf16 cell {hi x 4, lo x 4}: hi = bf16(v), lo = bf16(v - hi)
__device__ __forceinline__ u32x4 split_bf16(const f32x4 v) {
    const uint32_t h01 = pack_bf16(v[0], v[1]), h23 = pack_bf16(v[2], v[3]);
    const float r0 = v[0] - __uint_as_float(h01 << 16), r1 = v[1] - __uint_as_float(h01 & 0xFFFF0000u);
    const float r2 = v[2] - __uint_as_float(h23 << 16), r3 = v[3] - __uint_as_float(h23 & 0xFFFF0000u);
    return u32x4{h01, h23, pack_bf16(r0, r1), pack_bf16(r2, r3)};
}

// Pool-window max of one lane: columns 3*wi .. 3*wi+2 of its 12 columns, rows t = 0..2.
// Column c12 lives in tile j = c12 >> 2, register r = c12 & 3.
template <int WI>
__device__ __forceinline__ float window_max(const f32x4 (&acc)[3][3][3], int nt) {
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 3 * WI; c < 3 * WI + 3; ++c) m = fmaxf(m, acc[t][c >> 2][nt][c & 3]);
    return m;
}

// ReLU of the 3x3 max of pool window WI over single-group accumulators, as a signed-integer max of
// the bit patterns with 0: every negative float is a negative integer, non-negative floats order
// like their bit patterns, so max(0, bits...) is exactly relu(max(...)).  Integer maxima fuse into
// v_max3_i32 (4 + 1 instructions per window); the float form costs the compiler an extra
// NaN-quieting instruction per operand, and each of these is a VALU slot the MFMAs cannot use.
// (Inline-asm v_max3_f32 is not an option: the compiler does not guard inline asm against the
// MFMA-result read hazard.)
template <int WI>
__device__ __forceinline__ float window_max(const f32x4 (&acc)[3][3]) {
    int m = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 3 * WI; c < 3 * WI + 3; ++c) m = max(m, __float_as_int(acc[t][c >> 2][c & 3]));
    return __int_as_float(m);
}

// ------------------------------------------------------------------------------------------
// Block 1: 3 -> 48 channels, K = 27 (padded to 28 = 7 k-steps).
// One workgroup (4 waves) = 3 pooled rows of one frame = 255 pool windows = 16 wave tiles
// (windows enumerated row-major, 16 per tile; only the 256th is padding).  The 11 input rows
// x 258 columns x 3 channels the tile needs are staged in LDS as fp32, either copied from the
// normalised NCHW tensor (MODE 0) or produced on the fly from the u8 source frame by the
// resize (MODE 1: pure point sample, MODE 2: fixed-point bilinear / box), so the resized frame
// never exists in HBM.
// ------------------------------------------------------------------------------------------
constexpr int kC1Tile = kInC * kC1Rows * kC1RS;

// frame / tile of workgroup b when `per_frame` workgroups make up a frame (XCD-aware order)
__device__ __forceinline__ void xcd_frame_tile(int b, int per_frame, int& frame, int& tile) {
    const int xcd = b & 7, i = b >> 3;
    frame = xcd + 8 * (i / per_frame);
    tile = i % per_frame;
}

// 4-byte load at any byte address (the hardware handles unaligned global loads)
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

#ifndef CD_DMA_AUX
#define CD_DMA_AUX 0     // cache-policy bits of block 2's patch copies (global_load ... lds)
#endif
#ifndef CD_DIAG_P
#define CD_DIAG_P 0      // diagnostic builds: the arithmetic (0 fp32, 1 bf16, 2 split bf16) whose blocks 2 / 3 write time stamps
#endif
#ifndef CD_C1_CPW
#define CD_C1_CPW 1
#endif
constexpr int kC1MaxChunksPerWG = CD_C1_CPW;   // consecutive 3-pool-row chunks one workgroup walks

// Correctly rounded k / 255 for k = 0..255 without a table: 1/255 as a two-term constant, the fma
// rounds k * (hi + lo) once (checked exhaustively, in exact arithmetic, against the division
// data.py:227 performs).
__device__ __forceinline__ float byte_to_unit(float k) {
    const float hi = 0x1.010102p-8f, lo = -0x1.fdfdfep-33f;     // hi = fl(1/255), lo = fl(1/255 - hi)
    return fmaf(k, hi, k * lo);
}

// Staging helpers -------------------------------------------------------------------------
// Thread t owns LDS column t (network column t-1) for all 11 rows; threads 0..10 also own one
// cell of column 256 (row = t).

// MODE 1 fetch: one unaligned dword per cell (its low 3 bytes, after a shift, are the pixel).
// Nothing here waits for a load: the values are first touched in point_commit().  Piece u of a
// chunk is row u of the thread's column (u < 11) or the thread's cell of column 256 (u = 11), so
// that a chunk's fetch can be dealt out over the 12 (wave tile, channel group) units of the
// previous chunk's MFMA work.
struct PointRaw { uint32_t v[kC1Rows + 1]; };
struct PointCol { int a, sh, a2, sh2; bool xin; };

__device__ __forceinline__ PointCol point_columns(const SampleArgs& sa, int tid) {
    PointCol c;
    const int last = sa.W0 * 3 - 4;            // last byte offset a 4-byte load may start at
    const int gx = tid - 1;
    c.xin = gx >= 0;
    const int xo = c.xin ? (sa.affine ? sa.xm * gx + sa.xa : sa.sx[gx]) * 3 : 0;
    c.a = min(xo, last);
    c.sh = (xo - c.a) * 8;
    const int xo2 = (sa.affine ? sa.xm * (kNetW - 1) + sa.xa : sa.sx[kNetW - 1]) * 3;
    c.a2 = min(xo2, last);
    c.sh2 = (xo2 - c.a2) * 8;
    return c;
}

__device__ __forceinline__ uint32_t point_fetch_piece(const uint8_t* __restrict__ fsrc, const SampleArgs& sa,
                                                      const PointCol& c, int gy0, int H, int tid, int u) {
    const int r = u < kC1Rows ? u : tid;
    const int gy = gy0 + r, gyc = min(max(gy, 0), H - 1);
    const uint8_t* p = fsrc + (int64_t)(sa.affine ? sa.ym * gyc + sa.ya : sa.sy[gyc]) * sa.row_stride +
                       (u < kC1Rows ? c.a : c.a2);
    const bool ok = gy >= 0 && gy < H && (u < kC1Rows ? c.xin : tid < kC1Rows);
    return ok ? *reinterpret_cast<const u32_unaligned*>(p) : 0u;
}

__device__ __forceinline__ void put_pixel(float* tile, int r, int c, uint32_t px, bool bgr) {
    const float f0 = byte_to_unit((float)(px & 255u)), f1 = byte_to_unit((float)((px >> 8) & 255u)),
                f2 = byte_to_unit((float)((px >> 16) & 255u));
    tile[(0 * kC1Rows + r) * kC1RS + c] = bgr ? f2 : f0;     // R plane (0 for padding cells)
    tile[(1 * kC1Rows + r) * kC1RS + c] = f1;
    tile[(2 * kC1Rows + r) * kC1RS + c] = bgr ? f0 : f2;
}

// bf16 tile of the bf16 block-1 kernel: one 8-byte cell per pixel, {R, G, B, 0} in bfloat16
constexpr int kC1RSB = 260;       // its row stride in pixels (columns -1 .. 256 used)
__device__ __forceinline__ void put_pixel(uint2* tile, int r, int c, uint32_t px, bool bgr) {
    const float f0 = byte_to_unit((float)(px & 255u)), f1 = byte_to_unit((float)((px >> 8) & 255u)),
                f2 = byte_to_unit((float)((px >> 16) & 255u));
    tile[r * kC1RSB + c] = uint2{pack_bf16(bgr ? f2 : f0, f1), pack_bf16(bgr ? f0 : f2, 0.f)};
}

// Tile of the split-bf16 block-1 kernel: one 16-byte cell per pixel, {R, G, B, 0, R, G, B, 0} with the
// byte values themselves (0..255 are exact in bfloat16; the 1/255 is folded into the weights).
__device__ __forceinline__ void put_pixel(u32x4* tile, int r, int c, uint32_t px, bool bgr) {
    const float f0 = (float)(px & 255u), f1 = (float)((px >> 8) & 255u), f2 = (float)((px >> 16) & 255u);
    const uint32_t rg = pack_bf16(bgr ? f2 : f0, f1), b0 = pack_bf16(bgr ? f0 : f2, 0.f);
    tile[r * kC1RSB + c] = u32x4{rg, b0, rg, b0};
}

template <typename T>
__device__ __forceinline__ void point_commit(T* tile, int tid, PointRaw& raw, const PointCol& c, bool bgr) {
    // (each row's conversion is pinned behind the previous row's LDS writes: scheduled freely, the
    // conversions of all 36 values are hoisted to where the loads were issued and spill)
#pragma unroll
    for (int r = 0; r < kC1Rows; ++r) {
        asm volatile("" : "+v"(raw.v[r]));
        put_pixel(tile, r, tid, raw.v[r] >> c.sh, bgr);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(raw.v[kC1Rows]));
    if (tid < kC1Rows) put_pixel(tile, tid, 256, raw.v[kC1Rows] >> c.sh2, bgr);
}

// MODE 3: general fixed-point bilinear / 2x2 box straight from global memory (the fallback for
// source rows that are not 16-byte aligned or too wide to stage).  The two horizontally adjacent
// source pixels of a tap pair are 6 contiguous bytes, fetched as one unaligned 8-byte load per source
// row; all 22 loads of a column are in flight before the first is used.  Arithmetic as sample_u8().
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;

template <typename T>
__device__ __forceinline__ void generic_stage(T* tile, const uint8_t* __restrict__ fsrc,
                                              const SampleArgs& sa, int gy0, int H, int tid, bool bgr) {
    const int last = sa.W0 * 3 - 8;            // last byte offset an 8-byte load may start at
    for (int c = tid; c < 257; c += 256) {
        const int gx = c - 1;
        const bool xin = gx >= 0;
        const int gxc = max(gx, 0);
        const int c0 = sa.mode == 2 ? 2 * gxc : sa.sx[gxc];
        const int a0 = sa.mode == 2 ? 1 : sa.ax[2 * gxc], a1 = sa.mode == 2 ? 1 : sa.ax[2 * gxc + 1];
        const int start = min(c0 * 3, last);
        const int sh0 = (c0 * 3 - start) * 8;                  // bit offset of pixel c0 in the 8 bytes
        const int sh1 = c0 + 1 < sa.W0 ? sh0 + 24 : sh0;       // pixel c0+1 (or c0 again at the border)
        uint64_t top[kC1Rows], bot[kC1Rows];
#pragma unroll
        for (int r = 0; r < kC1Rows; ++r) {
            const int gy = min(max(gy0 + r, 0), H - 1);
            int r0, r1;
            if (sa.mode == 2) { r0 = 2 * gy; r1 = 2 * gy + 1; }
            else { const int sy = sa.sy[gy]; r0 = min(max(sy, 0), sa.H0 - 1); r1 = min(max(sy + 1, 0), sa.H0 - 1); }
            top[r] = *reinterpret_cast<const u64_unaligned*>(fsrc + (int64_t)r0 * sa.row_stride + start);
            bot[r] = *reinterpret_cast<const u64_unaligned*>(fsrc + (int64_t)r1 * sa.row_stride + start);
        }
#pragma unroll
        for (int r = 0; r < kC1Rows; ++r) {
            const int gy = gy0 + r;
            uint32_t px = 0;
            if (xin && gy >= 0 && gy < H) {
                const int b0 = sa.mode == 2 ? 1 : sa.by[2 * gy], b1 = sa.mode == 2 ? 1 : sa.by[2 * gy + 1];
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const int p00 = (int)((top[r] >> (sh0 + 8 * cc)) & 255), p01 = (int)((top[r] >> (sh1 + 8 * cc)) & 255);
                    const int p10 = (int)((bot[r] >> (sh0 + 8 * cc)) & 255), p11 = (int)((bot[r] >> (sh1 + 8 * cc)) & 255);
                    int v;
                    if (sa.mode == 2) v = (p00 + p01 + p10 + p11 + 2) >> 2;
                    else {
                        const int S0 = p00 * a0 + p01 * a1, S1 = p10 * a0 + p11 * a1;
                        v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
                        v = min(max(v, 0), 255);
                    }
                    px |= (uint32_t)v << (8 * cc);
                }
            }
            put_pixel(tile, r, c, px, bgr);
        }
    }
}

// MODE 2: the same arithmetic with coalesced loads.  For each of the tile's 11 rows the workgroup
// copies the two source rows it blends into LDS with 16-byte loads (every byte of a row is fetched
// once, by consecutive lanes), then each thread picks its two horizontally adjacent source pixels -
// 6 contiguous bytes - out of LDS as three aligned dwords per source row.  `stage` holds two rows of
// `pitch` bytes (a multiple of 16, at least 16 beyond the row).  Needs 16-byte aligned source rows.
template <typename T>
__device__ __forceinline__ void staged_rows_stage(T* tile, uint8_t* stage, int pitch, const uint8_t* __restrict__ fsrc,
                                                  const SampleArgs& sa, int gy0, int H, int tid, bool bgr) {
    const int chunks = (sa.W0 * 3 + 15) >> 4;            // 16-byte pieces per source row (the tail piece reads
                                                         // into the next row or the slack the caller guarantees)
    // this thread's column(s): LDS column tid (network column tid - 1) and, for tid 0, column 256 as well
    int c0s[2], a0s[2], a1s[2], offs[2], sh0s[2], sh1s[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = k == 0 ? tid : 256, gxc = max(c - 1, 0);
        c0s[k] = sa.mode == 2 ? 2 * gxc : sa.sx[gxc];
        a0s[k] = sa.mode == 2 ? 1 : sa.ax[2 * gxc];
        a1s[k] = sa.mode == 2 ? 1 : sa.ax[2 * gxc + 1];
        const int byte0 = c0s[k] * 3;
        offs[k] = byte0 & ~3;                              // aligned dword that holds the first byte
        sh0s[k] = (byte0 & 3) * 8;                         // bit offset of pixel c0 in the 12 bytes read
        sh1s[k] = c0s[k] + 1 < sa.W0 ? sh0s[k] + 24 : sh0s[k];   // pixel c0 + 1 (c0 again at the right border)
    }
    for (int r = 0; r < kC1Rows; ++r) {
        const int gy = gy0 + r;
        const bool yin = gy >= 0 && gy < H;                // uniform over the workgroup
        int b0 = 1, b1 = 1;
        if (yin) {
            int r0, r1;
            if (sa.mode == 2) { r0 = 2 * gy; r1 = 2 * gy + 1; }
            else {
                const int sy = sa.sy[gy];
                r0 = min(max(sy, 0), sa.H0 - 1); r1 = min(max(sy + 1, 0), sa.H0 - 1);
                b0 = sa.by[2 * gy]; b1 = sa.by[2 * gy + 1];
            }
            const uint4* g0 = reinterpret_cast<const uint4*>(fsrc + (int64_t)r0 * sa.row_stride);
            const uint4* g1 = reinterpret_cast<const uint4*>(fsrc + (int64_t)r1 * sa.row_stride);
            __syncthreads();                               // the previous row's taps are done with the staging rows
            for (int c = tid; c < chunks; c += 256) {
                reinterpret_cast<uint4*>(stage)[c] = g0[c];
                reinterpret_cast<uint4*>(stage + pitch)[c] = g1[c];
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && tid != 0) break;
            const int c = k == 0 ? tid : 256;
            uint32_t px = 0;
            if (yin && c >= 1) {
                const uint32_t* t = reinterpret_cast<const uint32_t*>(stage + offs[k]);
                const uint32_t* bt = reinterpret_cast<const uint32_t*>(stage + pitch + offs[k]);
                const uint64_t tlo = t[0] | ((uint64_t)t[1] << 32), blo = bt[0] | ((uint64_t)bt[1] << 32);
                const uint32_t thi = t[2], bhi = bt[2];
                // 9 bytes starting at bit sh0 of the 12 read: pixel c0 = bits 0..23, pixel c0 + 1 = bits 24..47
                const uint64_t top = sh0s[k] ? (tlo >> sh0s[k]) | ((uint64_t)thi << (64 - sh0s[k])) : tlo;
                const uint64_t bot = sh0s[k] ? (blo >> sh0s[k]) | ((uint64_t)bhi << (64 - sh0s[k])) : blo;
                const int d = sh1s[k] - sh0s[k];           // 24, or 0 at the right border
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const int p00 = (int)((top >> (8 * cc)) & 255), p01 = (int)((top >> (d + 8 * cc)) & 255);
                    const int p10 = (int)((bot >> (8 * cc)) & 255), p11 = (int)((bot >> (d + 8 * cc)) & 255);
                    int v;
                    if (sa.mode == 2) v = (p00 + p01 + p10 + p11 + 2) >> 2;
                    else {
                        const int S0 = p00 * a0s[k] + p01 * a1s[k], S1 = p10 * a0s[k] + p11 * a1s[k];
                        v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
                        v = min(max(v, 0), 255);
                    }
                    px |= (uint32_t)v << (8 * cc);
                }
            }
            put_pixel(tile, r, c, px, bgr);
        }
    }
}

// MODE 0: normalised fp32 NCHW input, row-wise copy.
__device__ __forceinline__ void f32_stage(float* tile, const float* __restrict__ fx, int gy0, int H, int tid) {
    for (int idx = tid; idx < kInC * kC1Rows * 258; idx += 256) {
        int c = idx % 258, rr = idx / 258;
        int r = rr % kC1Rows, ci = rr / kC1Rows;
        int gy = gy0 + r, gx = c - 1;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < kNetW) v = fx[((int64_t)ci * H + gy) * kNetW + gx];
        tile[(ci * kC1Rows + r) * kC1RS + c] = v;
    }
}

// MODE 0: fp32 NCHW input.  MODE 1: u8 source, pure point sample (every tap weight is 1).
// MODE 2: u8 source, general resize, source rows staged in LDS (dynamic shared memory: two rows).
// MODE 3: u8 source, general resize, per-thread gathers.
//
// A workgroup walks `cpw` consecutive chunks of one frame (one by default: with one chunk per
// workgroup the staging of one workgroup overlaps the MFMA phases of the three others on the CU, and
// that measured faster than walking several).  When it walks several (CD_C1_CPW > 1), the source
// bytes of chunk k+1 are requested (into registers) during the MFMA work of chunk k in MODE 1, one
// load instruction per (wave tile, channel group) unit, and converted and written to LDS after it.
#ifndef CD_C1_LB
#define CD_C1_LB 4
#endif
template <int MODE, int OBF>
__global__ __launch_bounds__(256, CD_C1_LB) void conv1_kernel(
    const float* __restrict__ x, const uint8_t* __restrict__ src, SampleArgs sa, int64_t frame_stride,
    int chan_order, const float* __restrict__ wfrag, typename Prec<OBF>::elem* __restrict__ out, int n, int H, int Hp,
    int chunks, int groups, int cpw, int stage_pitch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_stage[];   // MODE 2: two source rows
    typedef typename Prec<OBF>::elem elem;       // the arithmetic is fp32 either way; OBF is the storage format
    __shared__ __attribute__((aligned(16))) float tile[kC1Tile];
    __shared__ __attribute__((aligned(16))) float scratch[4 * 16 * 16];   // per-wave 16x16 transposes
    __shared__ float spost[3 * kC];               // bias / BN scale / BN shift (registers are scarcer)
    const int tid = threadIdx.x;
    int frame, grp;
    xcd_frame_tile(blockIdx.x, groups, frame, grp);
    if (frame >= n) return;
    const int chunk0 = grp * cpw;
    const int nchunks = min(cpw, chunks - chunk0);
    const bool bgr = chan_order == CD_BGR;
    const uint8_t* fsrc = MODE != 0 ? src + (int64_t)frame * frame_stride : nullptr;
    const float* fx = MODE == 0 ? x + (int64_t)frame * kInC * H * kNetW : nullptr;

#ifdef CD_DIAG
    // diagnostic build: per-workgroup time stamps go to the slack frame behind the batch
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(out + (int64_t)n * (Hp + 2) * kP1W * kC) + 16 * blockIdx.x;
    if (!OBF && tid == 0) {
        stamps[0] = __builtin_amdgcn_s_memrealtime();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[3] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    // The set-up and staging code is short, latency-bound and mostly VALU; the other waves on this
    // SIMD are issuing MFMAs back to back and, being older, win every arbitration.  Run it at raised
    // priority and drop back before the MFMA loop.
    __builtin_amdgcn_s_setprio(3);
    PointRaw raw;
    PointCol pcol{};
    if (MODE == 1) {
        pcol = point_columns(sa, tid);
#pragma unroll
        for (int u = 0; u < kC1Rows + 1; ++u) {
#ifdef CD_DIAG
            if (!OBF && tid == 0 && (u & 3) == 0) stamps[10 + (u >> 2)] = __builtin_amdgcn_s_memrealtime();
#endif
            raw.v[u] = point_fetch_piece(fsrc, sa, pcol, 9 * chunk0 - 1, H, tid, u);
        }
    }
#ifdef CD_DIAG
    if (!OBF && tid == 0) stamps[7] = __builtin_amdgcn_s_memrealtime();
#endif

    const int wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, kq = lane >> 4;
    int koff[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        int tap = c1_tap(4 * s + kq);      // the pad slot reads a valid cell; its weight is 0
        if (tap < 0) tap = -tap - 1;
        const int ci = tap / 100, ky = (tap / 10) % 10, kx = tap % 10;
        koff[s] = (ci * kC1Rows + ky) * kC1RS + kx;
    }
    const int q = lane >> 4, ch = lane & 15;
    elem* fout = out + (int64_t)frame * (Hp + 2) * kP1W * kC;
    float* my_out = scratch + wave * 256;
    // Everything the MFMA loop needs from global memory is fetched once, here, while the first tile
    // is being staged: a load issued inside the loop would cost a full memory round trip (about a
    // microsecond when the memory pipe is busy with other workgroups' gathers) per 63 MFMAs.
    // The host packs it as 16-byte pieces: 9 loads per lane, each a contiguous KB per wave.
    float bw[7][3];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(wfrag) + lane;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const f32x4 lo = wp[(2 * nt) * 64], hi = wp[(2 * nt + 1) * 64];
            bw[0][nt] = lo[0]; bw[1][nt] = lo[1]; bw[2][nt] = lo[2]; bw[3][nt] = lo[3];
            bw[4][nt] = hi[0]; bw[5][nt] = hi[1]; bw[6][nt] = hi[2];
        }
        if (tid < 3 * kC) {      // pieces 6..8 hold {bias, scale, shift, 0} per channel (lane & 15)
            const int c = tid % kC, which = tid / kC;
            spost[tid] = wfrag[((6 + c / 16) * 64 + (c & 15)) * 4 + which];
        }
    }

    for (int kc = 0; kc < nchunks; ++kc) {
        const int py0 = (chunk0 + kc) * kC1PoolRows;
        const int gy0 = 3 * py0 - 1;        // input row held in LDS row 0
        if (kc > 0) { __syncthreads(); __builtin_amdgcn_s_setprio(3); }   // every wave is done with the previous tile
#ifdef CD_DIAG
        if (!OBF && tid == 0 && kc == 0) stamps[4] = __builtin_amdgcn_s_memrealtime();
        if (!OBF && tid == 0 && kc == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
        if (MODE == 0) f32_stage(tile, fx, gy0, H, tid);
        else if (MODE == 1) point_commit(tile, tid, raw, pcol, bgr);
        else if (MODE == 2) staged_rows_stage(tile, dyn_stage, stage_pitch, fsrc, sa, gy0, H, tid, bgr);
        else generic_stage(tile, fsrc, sa, gy0, H, tid, bgr);
#ifdef CD_DIAG
        if (!OBF && tid == 0 && kc == 0) stamps[6] = __builtin_amdgcn_s_memrealtime();
#endif
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);
#ifdef CD_DIAG
        if (!OBF && tid == 0 && kc == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
#endif
        const bool more = MODE == 1 && kc + 1 < nchunks;

#pragma unroll
        for (int k = 0; k < 4; ++k) {            // wave tile wt = wave + 4k
            // (opaque to the compiler: it would otherwise hoist the address arithmetic of all four
            // tiles out of the chunk loop and keep ~100 registers of addresses alive)
            int wt = wave + 4 * k;
            asm volatile("" : "+v"(wt));
            // LDS offset (floats) of the pixel that A-row i of column group j maps to: window
            // W = 16wt + 4(i>>2) + (4j + (i&3))/3, column (4j + (i&3)) % 3 of that window
            int base[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c12 = 4 * j + (i & 3), d = (c12 * 11) >> 5, m = c12 - 3 * d;   // d = c12 / 3
                const int W = min(16 * wt + 4 * (i >> 2) + d, kC1Windows - 1);
                const int wr = (W >= kW1 ? 1 : 0) + (W >= 2 * kW1 ? 1 : 0);               // W / 85
                base[j] = 3 * wr * kC1RS + 3 * (W - wr * kW1) + m;
            }
            // this lane's piece of the transposed store: window lane>>2, channels 4*(lane&3)..+3
            const int Ws = 16 * wt + (lane >> 2);
            const int wrs = (Ws >= kW1 ? 1 : 0) + (Ws >= 2 * kW1 ? 1 : 0);
            const bool st_ok = Ws < kC1Windows && py0 + wrs < Hp;
            elem* st_ptr = fout + ((int64_t)(py0 + wrs + 1) * kP1W + (Ws - wrs * kW1) + 1) * kC + 4 * (lane & 3);
            // A-operand addresses of the 7 k-steps x 3 column groups (conv row t is an immediate)
            int addr[7][3];
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int j = 0; j < 3; ++j) addr[s][j] = base[j] + koff[s];
            // One 16-channel group at a time: 9 accumulators live instead of 27, which keeps the
            // kernel at 4 waves per SIMD so that other waves' MFMAs cover this wave's staging,
            // LDS latency and epilogue.  The A values are re-read from LDS for each group.
            // The three A values of the next conv row (t) are read while the three MFMAs of the
            // current one run (two sets of three registers; the scheduling barriers keep the compiler
            // from collapsing this back into read - wait - MFMA on a single register, which leaves a
            // wave issuing one MFMA per LDS round trip).  The pipeline runs across the channel
            // groups of a tile: row 0 of the next group has the same addresses as row 0 of this one.
            float a[2][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) a[0][j] = tile[addr[0][j]];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                if (kC1MaxChunksPerWG > 1 && more)
                    raw.v[k * 3 + nt] = point_fetch_piece(fsrc, sa, pcol, gy0 + 9, H, tid, k * 3 + nt);
                // Accumulators start from the conv bias (every register of a lane is one channel): the
                // first MFMA of each takes one shared {bias x 4} tuple as its C operand, which costs
                // nothing, where initialising 36 registers costs 36 VALU slots the MFMAs cannot use.
                const float bias = spost[nt * 16 + ch];
                const f32x4 bias4 = f32x4{bias, bias, bias, bias};
                f32x4 acc[3][3];
#pragma unroll
                for (int row = 0; row < 21; ++row) {
                    const int s = row / 3, t = row % 3;
                    const int cur = (nt * 21 + row) & 1;
                    const int nrow = row < 20 ? row + 1 : 0, ns = nrow / 3, ntt = nrow % 3;
                    if (row < 20 || nt < 2) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) a[cur ^ 1][j] = tile[addr[ns][j] + ntt * kC1RS];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[t][j] = mfma4(a[cur][j], bw[s][nt], s == 0 ? bias4 : acc[t][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float sc = spost[kC + nt * 16 + ch], sh = spost[2 * kC + nt * 16 + ch];
                // epilogue: lane (q, ch) owns pool windows 4q .. 4q+3 of the tile for channel ch + 16 nt.
                // The 16 windows x 16 channels are transposed through 1 KB of LDS so that every lane
                // stores 16 contiguous bytes (4 channels of one window).
                float m[4] = {window_max<0>(acc), window_max<1>(acc), window_max<2>(acc), window_max<3>(acc)};
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) my_out[(4 * q + wi) * 16 + ch] = fmaf(m[wi], sc, sh);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(my_out + (lane >> 2) * 16 + 4 * (lane & 3));
                    if (st_ok) {
                        if constexpr (OBF == kBF16)
                            *reinterpret_cast<uint2*>(st_ptr + nt * 16) = uint2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                        else if constexpr (OBF == kS2)
                            *reinterpret_cast<u32x4*>(st_ptr + nt * 16) = split_bf16(v);
                        else
                            *reinterpret_cast<f32x4*>(st_ptr + nt * 16) = v;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read done before the next group rewrites
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#ifdef CD_DIAG
    __syncthreads();
    if (!OBF && tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------------------------------
// Block 1 in bf16 mode: the same workgroup / wave-tile / epilogue structure, on
// v_mfma_f32_16x16x16_bf16.  The LDS tile holds one 8-byte cell {R, G, B, 0} per pixel, so the A
// fragment of conv row ky - K slots 4kq + e = (kx = kq, channel e), kq = 3 and e = 3 carrying zero
// weights - is one 8-byte LDS read at pixel (row + ky, col + kq): K = 27 goes through three MFMAs
// instead of seven, with no packing instructions.  The 15 fragments of a wave tile (5 tile rows x 3
// column groups) are read once and serve all 81 of its MFMAs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void f32_stage_bf16(uint2* tile, const float* __restrict__ fx, int gy0, int H, int tid) {
    for (int idx = tid; idx < kC1Rows * 258; idx += 256) {
        const int c = idx % 258, r = idx / 258;
        const int gy = gy0 + r, gx = c - 1;
        float v[3] = {0.f, 0.f, 0.f};
        if (gy >= 0 && gy < H && gx >= 0 && gx < kNetW)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) v[ci] = fx[((int64_t)ci * H + gy) * kNetW + gx];
        tile[r * kC1RSB + c] = uint2{pack_bf16(v[0], v[1]), pack_bf16(v[2], 0.f)};
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void conv1_bf16_kernel(
    const float* __restrict__ x, const uint8_t* __restrict__ src, SampleArgs sa, int64_t frame_stride,
    int chan_order, const bf16x4* __restrict__ wfrag, const float* __restrict__ post, uint16_t* __restrict__ out,
    int n, int H, int Hp, int chunks, int stage_pitch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_stage[];   // MODE 2: two source rows
    __shared__ __attribute__((aligned(16))) uint2 tile[kC1Rows * kC1RSB];
    __shared__ __attribute__((aligned(16))) float scratch[4 * 16 * 16];   // per-wave 16x16 transposes
    __shared__ float spost[3 * kC];
    const int tid = threadIdx.x;
    int frame, chunk;
    xcd_frame_tile(blockIdx.x, chunks, frame, chunk);
    if (frame >= n) return;
    const bool bgr = chan_order == CD_BGR;
    const uint8_t* fsrc = MODE != 0 ? src + (int64_t)frame * frame_stride : nullptr;
    const float* fx = MODE == 0 ? x + (int64_t)frame * kInC * H * kNetW : nullptr;
    const int py0 = chunk * kC1PoolRows;
    const int gy0 = 3 * py0 - 1;            // input row held in tile row 0

    PointRaw raw;
    PointCol pcol{};
    if (MODE == 1) {
        pcol = point_columns(sa, tid);
#pragma unroll
        for (int u = 0; u < kC1Rows + 1; ++u) raw.v[u] = point_fetch_piece(fsrc, sa, pcol, gy0, H, tid, u);
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int q = lane >> 4, ch = lane & 15;
    uint16_t* fout = out + (int64_t)frame * (Hp + 2) * kP1W * kC;
    float* my_out = scratch + wave * 256;
    // B fragments (ky, n-tile) stay in registers; bias and BatchNorm terms go to LDS
    bf16x4 bw[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) bw[ky][nt] = wfrag[(ky * 3 + nt) * 64 + lane];
    if (tid < 3 * kC) spost[tid] = post[tid];

    if (MODE == 0) f32_stage_bf16(tile, fx, gy0, H, tid);
    else if (MODE == 1) point_commit(tile, tid, raw, pcol, bgr);
    else if (MODE == 2) staged_rows_stage(tile, dyn_stage, stage_pitch, fsrc, sa, gy0, H, tid, bgr);
    else generic_stage(tile, fsrc, sa, gy0, H, tid, bgr);
    __syncthreads();

#pragma unroll
    for (int k = 0; k < 4; ++k) {            // wave tile wt = wave + 4k
        int wt = wave + 4 * k;
        asm volatile("" : "+v"(wt));         // keeps the compiler from hoisting all four tiles' addresses
        // tile cell of the pixel that A-row i of column group j maps to, shifted by this lane's kx
        int base[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c12 = 4 * j + (i & 3), d = (c12 * 11) >> 5, m = c12 - 3 * d;   // d = c12 / 3
            const int W = min(16 * wt + 4 * (i >> 2) + d, kC1Windows - 1);
            const int wr = (W >= kW1 ? 1 : 0) + (W >= 2 * kW1 ? 1 : 0);               // W / 85
            // (lanes kq = 3 carry zero weights; they re-read the kq = 2 cell, which is always written:
            // a never-written LDS cell may hold a NaN pattern, and NaN x 0 would poison the sum)
            base[j] = 3 * wr * kC1RSB + 3 * (W - wr * kW1) + m + min(kq, 2);
        }
        const int Ws = 16 * wt + (lane >> 2);
        const int wrs = (Ws >= kW1 ? 1 : 0) + (Ws >= 2 * kW1 ? 1 : 0);
        const bool st_ok = Ws < kC1Windows && py0 + wrs < Hp;
        uint16_t* st_ptr = fout + ((int64_t)(py0 + wrs + 1) * kP1W + (Ws - wrs * kW1) + 1) * kC + 4 * (lane & 3);
        bf16x4 a[5][3];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) a[r][j] = __builtin_bit_cast(bf16x4, tile[base[j] + r * kC1RSB]);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const float bias = spost[nt * 16 + ch];
            const f32x4 bias4 = f32x4{bias, bias, bias, bias};
            f32x4 acc[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[t + ky][j], bw[ky][nt],
                                                                               ky == 0 ? bias4 : acc[t][j], 0, 0, 0);
            const float sc = spost[kC + nt * 16 + ch], sh = spost[2 * kC + nt * 16 + ch];
            float m[4] = {window_max<0>(acc), window_max<1>(acc), window_max<2>(acc), window_max<3>(acc)};
#pragma unroll
            for (int wi = 0; wi < 4; ++wi) my_out[(4 * q + wi) * 16 + ch] = fmaf(m[wi], sc, sh);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            {
                const f32x4 v = *reinterpret_cast<const f32x4*>(my_out + (lane >> 2) * 16 + 4 * (lane & 3));
                if (st_ok) *reinterpret_cast<uint2*>(st_ptr + nt * 16) = uint2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read done before the next group rewrites
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------
// Block 1 in split-bf16 mode (u8 sources only): the workgroup / wave-tile / epilogue structure of the
// bf16 kernel on v_mfma_f32_16x16x32_bf16.  The input needs no splitting - a byte value is exact in
// bfloat16 - so the tile holds the bytes as they are and the weights carry the 1/255: w' = w / 255 in
// fp32, split into hi + lo bfloat16.  A lane's A operand is the pixel's 16-byte cell
// {R, G, B, 0, R, G, B, 0} at (row + ky, col + kq); its B operand is {w'_hi(R, G, B), 0, w'_lo(R, G, B), 0}
// of tap (ky, kx = kq) (zero for kq = 3): one MFMA per kernel row adds byte x (w'_hi + w'_lo), and
// every product is exact in the fp32 accumulator.  K = 27 in three MFMAs of 16 cycles each, where
// fp32 takes seven of 32.  Output cells are split bf16 like the other blocks' inputs (TERMS = 2).
//
// TERMS = 3 is block 1 of the fp32 arithmetic on u8 sources: a third bfloat16 term makes hi + lo + lo2 the
// fp32 weight w' EXACTLY (3 x 8 mantissa bits), every byte x term product is exact in the fp32 accumulator,
// and the activations are written as plain fp32 for the fp32 blocks 2 and 3.  Nothing is rounded that the
// fp32 matrix instructions would not round (they, too, add exact products into fp32 accumulators; only the
// order of the additions differs), at 5 MFMAs of 16 cycles per tile instead of 7 of 32: a pixel's cell has 8 K
// slots, {R, G, B, 0} twice, which meet {hi, lo} in one MFMA per kernel row; the third term needs the pixel once,
// so two kernel rows' pixels share an operand there (rows 0 + 1 in one MFMA, row 2 in another).
// ------------------------------------------------------------------------------------------
template <int MODE, int TERMS>
__global__ __launch_bounds__(256, TERMS == 3 ? 2 : 3) void conv1_split_kernel(
    const uint8_t* __restrict__ src, SampleArgs sa, int64_t frame_stride, int chan_order,
    const u32x4* __restrict__ wfrag, const u32x4* __restrict__ wfrag3, const float* __restrict__ post,
    uint32_t* __restrict__ out, int n, int H, int Hp, int chunks, int stage_pitch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_stage[];   // MODE 2: two source rows
    __shared__ __attribute__((aligned(16))) u32x4 tile[kC1Rows * kC1RSB];
    __shared__ __attribute__((aligned(16))) float scratch[4 * 16 * 16];   // per-wave 16x16 transposes
    __shared__ float spost[3 * kC];
    const int tid = threadIdx.x;
    int frame, chunk;
    xcd_frame_tile(blockIdx.x, chunks, frame, chunk);
    if (frame >= n) return;
    const bool bgr = chan_order == CD_BGR;
    const uint8_t* fsrc = src + (int64_t)frame * frame_stride;
    const int py0 = chunk * kC1PoolRows;
    const int gy0 = 3 * py0 - 1;            // input row held in tile row 0

    PointRaw raw;
    PointCol pcol{};
    if (MODE == 1) {
        pcol = point_columns(sa, tid);
#pragma unroll
        for (int u = 0; u < kC1Rows + 1; ++u) raw.v[u] = point_fetch_piece(fsrc, sa, pcol, gy0, H, tid, u);
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int q = lane >> 4, ch = lane & 15;
    uint32_t* fout = out + (int64_t)frame * (Hp + 2) * kP1W * kC;
    float* my_out = scratch + wave * 256;
    // B fragments (ky, n-tile) stay in registers; bias and BatchNorm terms go to LDS
    // third term: fragment 0 = {lo2 of kernel row 0, lo2 of kernel row 1}, fragment 1 = {lo2 of kernel row 2, 0}
    u32x4 bw[3][3], bw3[TERMS == 3 ? 2 : 1][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            bw[ky][nt] = wfrag[(ky * 3 + nt) * 64 + lane];
            if (TERMS == 3 && ky < 2) bw3[ky][nt] = wfrag3[(ky * 3 + nt) * 64 + lane];
        }
    if (tid < 3 * kC) spost[tid] = post[tid];

    if (MODE == 1) point_commit(tile, tid, raw, pcol, bgr);
    else if (MODE == 2) staged_rows_stage(tile, dyn_stage, stage_pitch, fsrc, sa, gy0, H, tid, bgr);
    else generic_stage(tile, fsrc, sa, gy0, H, tid, bgr);
    __syncthreads();

#pragma unroll
    for (int k = 0; k < 4; ++k) {            // wave tile wt = wave + 4k
        int wt = wave + 4 * k;
        asm volatile("" : "+v"(wt));         // keeps the compiler from hoisting all four tiles' addresses
        int base[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c12 = 4 * j + (i & 3), d = (c12 * 11) >> 5, m = c12 - 3 * d;   // d = c12 / 3
            const int W = min(16 * wt + 4 * (i >> 2) + d, kC1Windows - 1);
            const int wr = (W >= kW1 ? 1 : 0) + (W >= 2 * kW1 ? 1 : 0);               // W / 85
            // (lanes kq = 3 carry zero weights and re-read the kq = 2 cell, which is always written)
            base[j] = 3 * wr * kC1RSB + 3 * (W - wr * kW1) + m + min(kq, 2);
        }
        const int Ws = 16 * wt + (lane >> 2);
        const int wrs = (Ws >= kW1 ? 1 : 0) + (Ws >= 2 * kW1 ? 1 : 0);
        const bool st_ok = Ws < kC1Windows && py0 + wrs < Hp;
        uint32_t* st_ptr = fout + ((int64_t)(py0 + wrs + 1) * kP1W + (Ws - wrs * kW1) + 1) * kC + 4 * (lane & 3);
        u32x4 a[5][3];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) a[r][j] = tile[base[j] + r * kC1RSB];
        // third term: the pixels of kernel rows 0 and 1 share an operand (each cell's second half only repeats
        // its first), so the term costs two MFMAs per tile, not three
        u32x4 a01[TERMS == 3 ? 3 : 1][3];
        if (TERMS == 3) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < 3; ++j) a01[t][j] = u32x4{a[t][j][0], a[t][j][1], a[t + 1][j][0], a[t + 1][j][1]};
        }
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const float bias = spost[nt * 16 + ch];
            const f32x4 bias4 = f32x4{bias, bias, bias, bias};
            f32x4 acc[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[t + ky][j]),
                                                                            __builtin_bit_cast(bf16x8, bw[ky][nt]),
                                                                            ky == 0 ? bias4 : acc[t][j], 0, 0, 0);
            if (TERMS == 3) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a01[t][j]),
                                                                            __builtin_bit_cast(bf16x8, bw3[0][nt]),
                                                                            acc[t][j], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[t + 2][j]),
                                                                            __builtin_bit_cast(bf16x8, bw3[1][nt]),
                                                                            acc[t][j], 0, 0, 0);
            }
            const float sc = spost[kC + nt * 16 + ch], sh = spost[2 * kC + nt * 16 + ch];
            float m[4] = {window_max<0>(acc), window_max<1>(acc), window_max<2>(acc), window_max<3>(acc)};
#pragma unroll
            for (int wi = 0; wi < 4; ++wi) my_out[(4 * q + wi) * 16 + ch] = fmaf(m[wi], sc, sh);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            {
                const f32x4 v = *reinterpret_cast<const f32x4*>(my_out + (lane >> 2) * 16 + 4 * (lane & 3));
                if (st_ok) *reinterpret_cast<u32x4*>(st_ptr + nt * 16) = TERMS == 3 ? __builtin_bit_cast(u32x4, v) : split_bf16(v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read done before the next group rewrites
            __builtin_amdgcn_wave_barrier();
        }
    }
}

static unsigned xcd_grid(int n, int per_frame) { return (unsigned)(((n + 7) / 8) * 8 * per_frame); }

// Chunks per workgroup: as many as still leave four workgroups for every CU (the most that fit),
// at most kC1MaxChunksPerWG.
static int c1_chunks_per_wg(int n, int chunks) {
    const long total = (long)n * chunks;
    return (int)std::min<long>(kC1MaxChunksPerWG, std::max<long>(1, total / 1024));
}

// fp32 handles on u8 sources: block 1 as byte x three-term weights on the bf16 matrix pipe (exact, see
// conv1_split_kernel) unless CD_OPT_F32_BLOCK1_FP32_MFMA asks for the fp32 matrix instructions
static bool c1_three_terms(const cd_handle* h) { return h->dtype == CD_F32 && !h->opt[CD_OPT_F32_BLOCK1_FP32_MFMA]; }

// static LDS of the block-1 kernel a u8 forward of this handle launches
static int c1_static_lds(const cd_handle* h) {
    const int common = 4 * 16 * 16 * 4 + 3 * kC * 4;
    if (h->dtype == CD_BF16) return kC1Rows * kC1RSB * 8 + common;
    if (h->dtype == CD_BF16_SPLIT || c1_three_terms(h)) return kC1Rows * kC1RSB * 16 + common;
    return kC1Tile * 4 + common;
}

template <int MODE>
static void launch_conv1(const cd_handle* h, const Workspace& ws, unsigned grid, hipStream_t s, const float* x, const uint8_t* src,
                         const SampleArgs& sa, int64_t frame_stride, int chan_order, int n, const Geometry& g,
                         int chunks, int groups, int cpw, int stage_pitch = 0) {
    const size_t dyn = (size_t)2 * stage_pitch;
    if (h->dtype == CD_BF16)
        hipLaunchKernelGGL((conv1_bf16_kernel<MODE>), dim3(xcd_grid(n, chunks)), dim3(256), dyn, s, x, src, sa, frame_stride,
                           chan_order, reinterpret_cast<const bf16x4*>(h->p.w1b), (const float*)h->p.post,
                           reinterpret_cast<uint16_t*>(ws.act1), n, g.h, g.h1, chunks, stage_pitch);
    else if (h->dtype == CD_BF16_SPLIT) {
        if constexpr (MODE == 0)       // float frames are not exact in bfloat16: the fp32 block 1, split-bf16 output
            hipLaunchKernelGGL((conv1_kernel<MODE, kS2>), dim3(grid), dim3(256), dyn, s, x, src, sa, frame_stride, chan_order,
                               (const float*)h->p.w1, reinterpret_cast<uint32_t*>(ws.act1), n, g.h, g.h1, chunks, groups, cpw,
                               stage_pitch);
        else
            hipLaunchKernelGGL((conv1_split_kernel<MODE, 2>), dim3(xcd_grid(n, chunks)), dim3(256), dyn, s, src, sa, frame_stride,
                               chan_order, reinterpret_cast<const u32x4*>(h->p.w1s), reinterpret_cast<const u32x4*>(h->p.w1x),
                               (const float*)h->p.post, reinterpret_cast<uint32_t*>(ws.act1), n, g.h, g.h1, chunks, stage_pitch);
    }
    else if (MODE != 0 && c1_three_terms(h))
        hipLaunchKernelGGL((conv1_split_kernel<(MODE == 0 ? 1 : MODE), 3>), dim3(xcd_grid(n, chunks)), dim3(256), dyn, s, src, sa,
                           frame_stride, chan_order, reinterpret_cast<const u32x4*>(h->p.w1s),
                           reinterpret_cast<const u32x4*>(h->p.w1x), (const float*)h->p.post,
                           reinterpret_cast<uint32_t*>(ws.act1), n, g.h, g.h1, chunks, stage_pitch);
    else
        hipLaunchKernelGGL((conv1_kernel<MODE, kF32>), dim3(grid), dim3(256), dyn, s, x, src, sa, frame_stride, chan_order,
                           (const float*)h->p.w1, ws.act1, n, g.h, g.h1, chunks, groups, cpw, stage_pitch);
}

int launch_conv1_f32(const cd_handle* h, const Workspace& ws, const float* x, int n, const Geometry& g, hipStream_t s) {
    const int chunks = (g.h1 + kC1PoolRows - 1) / kC1PoolRows;
    SampleArgs sa{};
    const int cpw = c1_chunks_per_wg(n, chunks), groups = (chunks + cpw - 1) / cpw;
    launch_conv1<0>(h, ws, xcd_grid(n, groups), s, x, nullptr, sa, 0, 0, n, g, chunks, groups, cpw);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

int launch_conv1_u8(const cd_handle* h, const Workspace& ws, const ResizeTables& rt, const uint8_t* src, int n, int H0,
                    int W0, int64_t row_stride, int64_t frame_stride, int chan_order, const Geometry& g, hipStream_t s) {
    const int chunks = (g.h1 + kC1PoolRows - 1) / kC1PoolRows;
    SampleArgs sa{rt.sx, rt.ax, rt.sy, rt.by, rt.mode, H0, W0, row_stride,
                  rt.affine, rt.xm, rt.xa, rt.ym, rt.ya};
    const int cpw = c1_chunks_per_wg(n, chunks), groups = (chunks + cpw - 1) / cpw;
    if (W0 < 3) { set_error("cd_forward_u8: source frames must be at least 3 pixels wide"); return CD_EINVAL; }
    // General resize: source rows go through LDS when they are 16-byte aligned and two of them fit
    // beside the kernel's static LDS in 64 KB (sources up to 4096 pixels wide in fp32, 2500 in split
    // bf16).  The last 16-byte piece of a row may reach up to 15 bytes past its pixels: that is only
    // done when it stays inside the row's own row_stride bytes (tail_ok below: whole rows of
    // row_stride bytes are readable for every row, the last one included, which cd_forward_u8
    // documents); otherwise the per-thread gather variant runs, which never reads past a row's pixels.
    const int row_bytes = W0 * 3, pitch = ((row_bytes + 15) & ~15) + 16;
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 15) == 0;
    const bool tail_ok = (row_bytes & 15) == 0 || row_stride >= (int64_t)((row_bytes + 15) & ~15);
    if (rt.mode == 1) launch_conv1<1>(h, ws, xcd_grid(n, groups), s, nullptr, src, sa, frame_stride, chan_order, n, g, chunks, groups, cpw);
    else if (aligned && tail_ok && 2 * pitch <= 65536 - c1_static_lds(h))
        launch_conv1<2>(h, ws, xcd_grid(n, groups), s, nullptr, src, sa, frame_stride, chan_order, n, g, chunks, groups, cpw, pitch);
    else launch_conv1<3>(h, ws, xcd_grid(n, groups), s, nullptr, src, sa, frame_stride, chan_order, n, g, chunks, groups, cpw);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

// ------------------------------------------------------------------------------------------
// 48 -> 48 channel wave tile (blocks 2 and 3), K = 432 = 27 steps of (ky, kx, 16-channel group).
// `patch` is channel-last in LDS with pixel stride kC elements; a lane's A fragment of a step is
// one 16-byte (fp32) or 8-byte (bf16) LDS read at
//     patch[base[j] + ((t + ky) * pitch + kx) * kC + 16 g],   base[j] includes 4 * (lane >> 4),
// i.e. input channels 16 g + 4 kq .. + 3 (kq = lane >> 4); the packed weights use the same order.
// ------------------------------------------------------------------------------------------
// The 27 steps are fully unrolled and software-pipelined by hand: the operands of step s+1 (nine
// LDS reads, three weight loads that hit L1/L2) are requested during the MFMAs of step s and first
// used after them.  The scheduling barriers keep the compiler from sinking the loads down to
// their first use, which would expose the L2 latency once per step.
template <int PITCH, int BF>
__device__ __forceinline__ void conv48_load_step(const typename Prec<BF>::elem* patch, const int (&base)[3],
                                                 __amdgpu_buffer_rsrc_t wr, int lane, int s,
                                                 typename Prec<BF>::frag (&a)[3][3], typename Prec<BF>::bfrag (&b)[3]) {
    typedef typename Prec<BF>::frag frag;
    const int kk = s / 3, g = s % 3;
    const int off = ((kk / 3) * PITCH + (kk % 3)) * kC + 16 * g;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) b[nt] = load_frag<BF>(wr, lane, s * 3 + nt);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            a[t][j] = *reinterpret_cast<const frag*>(patch + base[j] + off + t * PITCH * kC);
}

// Steps S0 .. S1-1 of the 27 (a K range: the accumulators then hold a partial sum).
template <int PITCH, int BF, int S0 = 0, int S1 = 27>
__device__ __forceinline__ void conv48_wave_tile(const typename Prec<BF>::elem* patch, const int (&base)[3],
                                                 __amdgpu_buffer_rsrc_t wr, int lane, f32x4 (&acc)[3][3][3]) {
    typedef typename Prec<BF>::frag frag;
    constexpr int kM = 27 * Prec<BF>::kSub;      // MFMAs per step
    frag a[2][3][3];
    typename Prec<BF>::bfrag b[2][3];
    conv48_load_step<PITCH, BF>(patch, base, wr, lane, S0, a[S0 & 1], b[S0 & 1]);
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        const int cur = s & 1;
        const int nk = (s + 1) / 3, ng = (s + 1) % 3;
        const int off = ((nk / 3) * PITCH + (nk % 3)) * kC + 16 * ng;
        // next step's 3 weight loads and 9 LDS reads go out one per MFMA: with one wave per SIMD,
        // anything issued as a block between two MFMAs leaves the matrix pipe idle
        const int n_side = s + 1 < S1 ? 12 : 0;
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            const int e = m / 27, t = (m / 9) % 3, j = (m / 3) % 3, nt = m % 3;
            if (s == S0 && e == 0)
                acc[t][j][nt] = mfma_frag<BF>(a[cur][t][j], b[cur][nt], e, f32x4{0.f, 0.f, 0.f, 0.f});
            else
                acc[t][j][nt] = mfma_frag<BF>(a[cur][t][j], b[cur][nt], e, acc[t][j][nt]);
            if (m < n_side) {
                __builtin_amdgcn_sched_barrier(0);
                if (m < 3) {
                    b[cur ^ 1][m] = load_frag<BF>(wr, lane, (s + 1) * 3 + m);
                } else {
                    const int lt = (m - 3) / 3, lj = (m - 3) % 3;
                    a[cur ^ 1][lt][lj] = *reinterpret_cast<const frag*>(patch + base[lj] + off + lt * PITCH * kC);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------
// Block 2: persistent single-wave workgroups, four per CU (one wave per SIMD).
//
// A wave walks wave tiles (4x4 pool windows = 12x12 conv pixels of one frame).  The K loop runs
// channel-group outermost: for each 16-channel group the 14x14 input patch of that group
// (12.25 KB, [pixel][16 floats]) is brought in by LDS-DMA into one of two buffers while the
// previous group's 972 MFMAs run, so the copy never shows up in the wave's timeline.  The
// operand prefetch of conv48 (one step ahead) runs straight across group and tile boundaries.
// VMEM ops complete in issue order, so any wait the compiler places on a weight load that was
// issued after a DMA also covers that DMA; the explicit vmcnt(0) at each group boundary (where
// only loads issued a full step earlier are outstanding) makes that ordering unconditional.
// ------------------------------------------------------------------------------------------
#ifndef CD_C2_BF16_WAVES
#define CD_C2_BF16_WAVES 2048
#endif
constexpr int kP2 = 14;
constexpr int kG2Pix = kP2 * kP2;              // 196 pixels
// per precision: 16-byte pieces of one channel group of a patch (16 channels x element size per
// pixel), wave-wide DMA instructions that copy them, and the LDS buffer padded to whole instructions
template <int BF> struct G2 {
    static constexpr int kEs = Prec<BF>::kEs;
    static constexpr int kPieces = kG2Pix * kEs;            // 784 fp32, 392 bf16
    static constexpr int kDma = (kPieces + 63) / 64;        // 13 fp32, 7 bf16
    static constexpr int kBufElems = kDma * 1024 / kEs;
};

template <int BF> struct Conv2Tile { const char* src; typename Prec<BF>::elem* dst; int ty, tx; };

// LDS-DMA of one channel group of a patch: wave-wide 16-byte copies, LDS image linear.
// All 64 lanes take part in the last copy as well (no exec-masked region: a branch here makes the
// compiler shuttle all 108 accumulators between AGPRs and VGPRs once per unit): its surplus
// pieces read the first pixels of patch rows 14 and 15, which exist (next rows of the padded frame
// or the slack frame), and land in the padding at the end of the LDS buffer.
template <int BF>
__device__ __forceinline__ void conv2_dma_group(const char* src, typename Prec<BF>::elem* dst,
                                                const int (&rel)[G2<BF>::kDma], int lane) {
    char* d = reinterpret_cast<char*>(dst);
#pragma unroll
    for (int it = 0; it < G2<BF>::kDma; ++it)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint32_t)rel[it]),
                                         (__attribute__((address_space(3))) void*)(d + it * 1024), 16, 0, CD_DMA_AUX);
}

// One (tile, channel group) unit: 9 pipelined steps.  G, the channel group, is a compile-time
// constant and a tile always starts with operand set 0, so the set each of the 27 steps of a tile
// reads is fixed at compile time.  Which LDS buffer holds the unit's patch alternates from unit to
// unit and from tile to tile (a tile is an odd number of units); `mine_off` and `other_off`
// (elements) select them at run time.  The first MFMA of each accumulator in a tile takes a literal
// zero as its C operand, so the accumulators are defined and consumed inside one tile body and
// never cross the loop edge (the compiler otherwise permutes all 108 of them through VGPRs once
// per unit).
template <int G, int BF, int KBD>
__device__ __forceinline__ void conv2_unit(const typename Prec<BF>::elem* bufs, int mine_off, int other_off,
                                           const char* next_src, const int (&rel)[G2<BF>::kDma],
                                           const int (&base)[3], __amdgpu_buffer_rsrc_t wr, int lane,
                                           typename Prec<BF>::frag (&a)[2][3][3],
                                           typename Prec<BF>::bfrag (&b)[KBD + 1][3],
                                           f32x4 (&acc)[3][3][3], unsigned long long* st = nullptr) {
    typedef typename Prec<BF>::frag frag;
    typedef typename Prec<BF>::elem elem;
    constexpr int kM = 27 * Prec<BF>::kSub;      // MFMAs per step
    constexpr int kBD = KBD, kNB = kBD + 1;
    if (st && lane == 0) st[9] = __builtin_amdgcn_s_memrealtime();
    // The patch copy issued in step 0 of the previous unit must have landed.  VMEM completes in issue
    // order, so it is enough to wait for everything but the weight loads of the last kBD - 1 steps
    // (with one step of look-ahead: everything; all of it was issued at least one step ago).
    if constexpr (kBD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kBD - 1) * Prec<BF>::kBLoads) : "memory");
    int bm[3], bo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { bm[j] = base[j] + mine_off; bo[j] = base[j] + other_off; }
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) {
        constexpr int kFirst = G * 9;
        const int c = (kFirst + kk) & 1;
        // the weight fragments rotate through kBD + 1 register sets: this step's, and where the ones
        // requested now (for kBD steps ahead; a tile is 27 steps, whatever tile follows) go
        const int bc = kBD == 1 ? c : (kFirst + kk) % kNB, bn = kBD == 1 ? c ^ 1 : (kFirst + kk + kBD) % kNB;
        const int bstep = (kFirst + kk + kBD) % 27, bg = bstep / 9, bk = bstep % 9;
        // Next step's operands: from this unit's patch, or (last step) from the next unit's
        const int* ab = kk < 8 ? bm : bo;
        const int nk = kk < 8 ? kk + 1 : 0;
        const int off = ((nk / 3) * kP2 + (nk % 3)) * 16;
        // The wave is alone on its SIMD, so whatever it issues between two MFMAs that takes longer
        // than the MFMA leaves the matrix pipe idle.  The step's memory instructions are therefore
        // dealt out one per MFMA instead of going out as a block: 3 weight loads, 9 LDS reads and,
        // in step 0, the copies of the next unit's patch (behind the operand loads, so that the
        // wait for those - VMEM completes in order - does not have to cover the copies).
        if (st && lane == 0) st[kk] = __builtin_amdgcn_s_memrealtime();
        const int n_side = kk == 0 ? 12 + G2<BF>::kDma : 12;
        static_assert(12 + G2<BF>::kDma <= kM, "more memory instructions than MFMAs in a step");
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            const int e = m / 27, t = (m / 9) % 3, j = (m / 3) % 3, nt = m % 3;
            if (G == 0 && kk == 0 && e == 0)
                acc[t][j][nt] = mfma_frag<BF>(a[c][t][j], b[bc][nt], e, f32x4{0.f, 0.f, 0.f, 0.f});
            else
                acc[t][j][nt] = mfma_frag<BF>(a[c][t][j], b[bc][nt], e, acc[t][j][nt]);
            if (m < n_side) {
                __builtin_amdgcn_sched_barrier(0);
                if (m < 3) {
#ifdef CD_EXP_NOB
                    b[bn][m] = b[bc][m];
#else
                    b[bn][m] = load_frag<BF>(wr, lane, (bk * 3 + bg) * 3 + m);
#endif
                } else if (m < 12) {
                    const int lt = (m - 3) / 3, lj = (m - 3) % 3;
#ifdef CD_EXP_NOA
                    a[c ^ 1][lt][lj] = a[c][lt][lj];
#else
                    a[c ^ 1][lt][lj] = *reinterpret_cast<const frag*>(bufs + ab[lj] + off + lt * kP2 * 16);
#endif
                } else {
                    const int it = m - 12;
#ifndef CD_EXP_NODMA
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(next_src + (uint32_t)rel[it]),
                        (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(const_cast<elem*>(bufs) + other_off) + it * 1024),
                        16, 0, CD_DMA_AUX);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// pooled + normalised results of one wave tile -> HBM, transposed through `scratch` (3 KB of an
// idle patch buffer) so that each pool window's 48 channels leave as 16-byte stores
template <int BF>
__device__ __forceinline__ void conv2_epilogue(float* scratch, const f32x4 (&acc)[3][3][3],
                                               const float* post, const Conv2Tile<BF>& t, int Hp,
                                               int lane) {
    const int q = lane >> 4, ch = lane & 15;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int c = nt * 16 + ch;
        const float bias = post[c], sc = post[kC + c], sh = post[2 * kC + c];
        float m[4] = {window_max<0>(acc, nt), window_max<1>(acc, nt), window_max<2>(acc, nt),
                      window_max<3>(acc, nt)};
#pragma unroll
        for (int wi = 0; wi < 4; ++wi) scratch[(4 * q + wi) * kC + c] = fmaf(fmaxf(m[wi] + bias, 0.f), sc, sh);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if constexpr (BF == kBF16) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int f = it * 64 + lane;                 // 16-byte piece f (8 channels) of 16 windows x 96 bytes
            const int win = min(f / 6, 15), c8 = f % 6;
            const int py = t.ty * 4 + (win >> 2), px = t.tx * 4 + (win & 3);
            const f32x4 lo = *reinterpret_cast<const f32x4*>(scratch + win * kC + 8 * c8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(scratch + win * kC + 8 * c8 + 4);
            const u32x4 v = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
            if (f < 96 && py < Hp && px < kW2)
                *reinterpret_cast<u32x4*>(t.dst + ((int64_t)(py + 1) * kP2W + px + 1) * kC + 8 * c8) = v;
        }
    } else {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int f = it * 64 + lane;                 // 16-byte piece f of 16 windows x 192 bytes
            const int win = f / 12, c4 = f % 12;
            const int py = t.ty * 4 + (win >> 2), px = t.tx * 4 + (win & 3);
            const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + win * kC + 4 * c4);
            if (py < Hp && px < kW2) {
                if constexpr (BF == kS2)
                    *reinterpret_cast<u32x4*>(t.dst + ((int64_t)(py + 1) * kP2W + px + 1) * kC + 4 * c4) = split_bf16(v);
                else
                    *reinterpret_cast<f32x4*>(t.dst + ((int64_t)(py + 1) * kP2W + px + 1) * kC + 4 * c4) = v;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// KBD: how many steps ahead of their use the weight fragments are requested (KBD + 1 register sets).
template <int BF, int KBD = Prec<BF>::kBDist>
__global__ __launch_bounds__(64) void conv2_kernel(const typename Prec<BF>::elem* __restrict__ in,
                                                   typename Prec<BF>::elem* __restrict__ out,
                                                   const void* __restrict__ wfrag,
                                                   const float* __restrict__ post, int n, int Hin, int Hp,
                                                   int tiles_x, int tiles_y, int total_tiles) {
    typedef typename Prec<BF>::elem elem;
    typedef typename Prec<BF>::frag frag;
    constexpr int kEs = Prec<BF>::kEs, kDma = G2<BF>::kDma, kBuf = G2<BF>::kBufElems;
    __shared__ __attribute__((aligned(16))) elem bufs[2 * kBuf];
    __shared__ float spost[3 * kC];               // bias / BN scale / BN shift: read by every epilogue
    const int lane = threadIdx.x;
    const int per_frame = tiles_x * tiles_y;
    const int64_t in_frame = (int64_t)(Hin + 2) * kP1W * kC * kEs, out_frame = (int64_t)(Hp + 2) * kP2W * kC;

    // tile id -> is it a real tile of this batch? (the grid is rounded up to 8 frames)
    auto valid = [&](int id) {
        int frame, tt;
        xcd_frame_tile(id, per_frame, frame, tt);
        return id >= 0 && id < total_tiles && frame < n;
    };
    // tile id -> source patch origin / destination frame; invalid ids map to tile 0 so that the
    // last look-ahead copy of a wave stays in bounds (its data is never used)
    auto locate = [&](int id) {
        int frame = 0, tt = 0;
        if (valid(id)) xcd_frame_tile(id, per_frame, frame, tt);
        Conv2Tile<BF> t;
        t.ty = tt / tiles_x; t.tx = tt % tiles_x;
        t.src = reinterpret_cast<const char*>(in) + frame * in_frame + ((int64_t)(12 * t.ty) * kP1W + 12 * t.tx) * kC * kEs;
        t.dst = out + frame * out_frame;
        return t;
    };
    int id = blockIdx.x;
    if (!valid(id)) return;

    // per-lane byte offsets of this lane's DMA pieces relative to the patch origin, group 0
    int rel[kDma];
#pragma unroll
    for (int it = 0; it < kDma; ++it) {
        const int f = it * 64 + lane, pix = f / kEs, piece = f % kEs;    // kEs pieces per pixel
        rel[it] = ((pix / kP2) * kP1W + pix % kP2) * kC * kEs + piece * 16;
    }
    const int i = lane & 15, kq = lane >> 4;
    int base[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) base[j] = ((3 * (i >> 2)) * kP2 + 4 * j + (i & 3)) * 16 + 4 * kq;
    const __amdgpu_buffer_rsrc_t wr = weight_rsrc<BF>(wfrag);

#ifdef CD_DIAG
    // diagnostic build (fp32 only): time stamps of the first 400 waves go to the slack frame behind the batch
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(out + (int64_t)n * out_frame) + 32 * blockIdx.x;
    int nstamp = 0;
#define CD_STAMP() do { if (BF == CD_DIAG_P && lane == 0 && nstamp < 32 && blockIdx.x < 200) stamps[nstamp] = __builtin_amdgcn_s_memrealtime(); ++nstamp; } while (0)
#else
#define CD_STAMP() do { } while (0)
#endif
    CD_STAMP();
    Conv2Tile<BF> cur = locate(id);
    int next_id = valid(id + (int)gridDim.x) ? id + (int)gridDim.x : -1;
    Conv2Tile<BF> nxt = locate(next_id);
    // everything the first step needs from global memory is requested together: the patch copy,
    // step 0's weight fragments and the affine terms share one round trip
    constexpr int kBD = KBD;
    frag a[2][3][3];
    typename Prec<BF>::bfrag b[kBD + 1][3];
    f32x4 acc[3][3][3];
    conv2_dma_group<BF>(cur.src, bufs, rel, lane);
    // weight fragments of the first kBD steps (step s of a tile = kernel tap s % 9 of channel group s / 9)
#pragma unroll
    for (int s0 = 0; s0 < kBD; ++s0)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) b[s0][nt] = load_frag<BF>(wr, lane, ((s0 % 9) * 3 + s0 / 9) * 3 + nt);
    spost[lane] = post[lane];
    spost[64 + lane] = post[64 + lane];
    if (lane < 3 * kC - 128) spost[128 + lane] = post[128 + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) a[0][t][j] = *reinterpret_cast<const frag*>(bufs + base[j] + t * kP2 * 16);
    CD_STAMP();

    // One loop iteration = one wave tile = three units.  `first` is the buffer (as an element
    // offset) that holds group 0 of the current tile; the look-ahead copy in the last unit puts
    // group 0 of the next tile into the other one.
    constexpr int kGroupBytes = 16 * kEs;
    int first = 0;
    for (;;) {
        const int second = kBuf - first;
        // Pin step 0's operands in their registers: left to itself the compiler re-loads the
        // weights here instead of carrying them across the loop edge, and the wait for that
        // re-load would also wait (VMEM completes in order) for the patch copy issued below.
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(a[0][t][j]));
#pragma unroll
        for (int s0 = 0; s0 < kBD; ++s0)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                #ifdef CD_S2_WDUP
                if constexpr (BF == kS2) { asm volatile("" : "+v"(b[s0][nt].h)); asm volatile("" : "+v"(b[s0][nt].l)); }
#else
                if constexpr (BF == kS2) asm volatile("" : "+v"(b[s0][nt].v));
#endif
                else asm volatile("" : "+v"(b[s0][nt]));
            }
#ifdef CD_DIAG
        unsigned long long* stp = (BF == CD_DIAG_P && blockIdx.x < 200 && nstamp == 14) ? stamps + 6400 : nullptr;   // third tile
        conv2_unit<0, BF, KBD>(bufs, first, second, cur.src + kGroupBytes, rel, base, wr, lane, a, b, acc, stp);
        CD_STAMP();
        conv2_unit<1, BF, KBD>(bufs, second, first, cur.src + 2 * kGroupBytes, rel, base, wr, lane, a, b, acc, stp ? stp + 10 : nullptr);
        CD_STAMP();
        conv2_unit<2, BF, KBD>(bufs, first, second, nxt.src, rel, base, wr, lane, a, b, acc, stp ? stp + 20 : nullptr);
        CD_STAMP();
#else
        conv2_unit<0, BF, KBD>(bufs, first, second, cur.src + kGroupBytes, rel, base, wr, lane, a, b, acc);
        conv2_unit<1, BF, KBD>(bufs, second, first, cur.src + 2 * kGroupBytes, rel, base, wr, lane, a, b, acc);
        conv2_unit<2, BF, KBD>(bufs, first, second, nxt.src, rel, base, wr, lane, a, b, acc);
#endif
        // `first` was last read one step ago; the next tile's first unit copies into it only after
        // the epilogue's final LDS wait
        conv2_epilogue<BF>(reinterpret_cast<float*>(bufs + first), acc, spost, cur, Hp, lane);
        CD_STAMP();
        if (next_id < 0) break;
        id = next_id;
        cur = nxt;
        next_id = valid(id + (int)gridDim.x) ? id + (int)gridDim.x : -1;
        nxt = locate(next_id);
        first = second;
        // 27 steps leave the next step's operands in set 1; a tile starts from set 0
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 3; ++j) a[0][t][j] = a[1][t][j];
        // (weight sets rotate with period kBD + 1: two sets flip like the operands, three sets come
        // round exactly once per 27-step tile)
        static_assert(kBD == 1 || 27 % (kBD + 1) == 0, "weight register sets must realign at a tile boundary");
        if constexpr (kBD == 1) {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) b[0][nt] = b[1][nt];
        }
    }
    // the look-ahead copy and operand loads of a unit that will never run are still in flight
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

int launch_conv2_split2(const cd_handle* h, const Workspace& ws, int n, const Geometry& g, hipStream_t s);

// far_weights: split bf16 only - the instance that requests its weights kS2FarWeights steps ahead instead of
// kBDist.  Worth its 100 extra registers only while another stream saturates HBM (the range call's resize
// pass at 1080p: +2.9 %; without such a stream 0.5 % slower), so the caller says when.
constexpr int kS2FarWeights = 8;
int launch_conv2(const cd_handle* h, const Workspace& ws, int n, const Geometry& g, hipStream_t s, bool far_weights) {
    const int tiles_y = (g.h2 + 3) / 4, tiles_x = (kW2 + 3) / 4;
    const int total = (int)xcd_grid(n, tiles_x * tiles_y);
    // persistent single-wave workgroups: one per SIMD in fp32 (the wave needs 308 registers and keeps
    // the matrix pipe busy on its own), two per SIMD in bf16 (the MFMAs are 4x shorter, so a second
    // wave has to cover the first one's LDS and epilogue latencies)
    const int cap = h->dtype == CD_BF16 ? CD_C2_BF16_WAVES : 1024;
    const int grid = total < cap ? total : cap;
    if (h->dtype == CD_BF16)
        hipLaunchKernelGGL(conv2_kernel<kBF16>, dim3((unsigned)grid), dim3(64), 0, s,
                           reinterpret_cast<const uint16_t*>(ws.act1), reinterpret_cast<uint16_t*>(ws.act2),
                           (const void*)h->p.w2b, h->p.post + 3 * kC, n, g.h1, g.h2, tiles_x, tiles_y, total);
    else if (h->dtype == CD_BF16_SPLIT && h->opt[CD_OPT_SPLIT_TWO_WAVES])
        return launch_conv2_split2(h, ws, n, g, s);
    else if (h->dtype == CD_BF16_SPLIT && far_weights)
        hipLaunchKernelGGL((conv2_kernel<kS2, kS2FarWeights>), dim3((unsigned)grid), dim3(64), 0, s,
                           reinterpret_cast<const uint32_t*>(ws.act1), reinterpret_cast<uint32_t*>(ws.act2),
                           (const void*)h->p.w2s, h->p.post + 3 * kC, n, g.h1, g.h2, tiles_x, tiles_y, total);
    else if (h->dtype == CD_BF16_SPLIT)
        hipLaunchKernelGGL(conv2_kernel<kS2>, dim3((unsigned)grid), dim3(64), 0, s,
                           reinterpret_cast<const uint32_t*>(ws.act1), reinterpret_cast<uint32_t*>(ws.act2),
                           (const void*)h->p.w2s, h->p.post + 3 * kC, n, g.h1, g.h2, tiles_x, tiles_y, total);
    else
        hipLaunchKernelGGL(conv2_kernel<kF32>, dim3((unsigned)grid), dim3(64), 0, s, (const float*)ws.act1, ws.act2,
                           (const void*)h->p.w2, h->p.post + 3 * kC, n, g.h1, g.h2, tiles_x, tiles_y, total);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

// ------------------------------------------------------------------------------------------
// Blocks 1 + 2 fused (split bf16): the block-1 activations never reach HBM.
//
// Same persistent single-wave workgroups and the same wave tile as conv2_kernel (4x4 pool windows of
// block 2 = a 14x14 patch of block-1 activations x 48 channels), but the patch is not copied from
// HBM: the wave computes it.  Its input is the 44x44 neighbourhood of the resized frame, held in LDS
// as the 8-byte operand cells {R, G, B, 0} (bfloat16 byte values) that the resize pass wrote to HBM
// - a plain strided copy, so it arrives by LDS-DMA while the previous tile's last unit runs.  Per
// 16-channel group g the wave then alternates
//     produce(g): 14 patch rows x (27 MFMAs of K = 32: three kernel rows x nine conv pixels-groups,
//                 the arithmetic of conv1_split_kernel; ReLU + 3x3 max + affine in registers; the
//                 results written to LDS as the {hi x 4, lo x 4} cells block 2 reads; pixels outside
//                 the frame are block 2's zero padding and are written as zeros)
//     consume(g): the 9 steps x 54 MFMAs of conv2_unit on that patch.
// The halo of a tile is recomputed ((14/12)^2 = 1.36 x the block-1 MFMAs), which is cheap next to the
// 835 KB per frame of activations written and read back otherwise.  LDS per wave: 15.1 KB of input
// cells + two 12.25 KB patch buffers = 40 576 bytes, four waves per CU.
// Values are bit-identical to conv1_split_kernel followed by conv2_kernel<kS2>.
// ------------------------------------------------------------------------------------------
constexpr int kFIn = 44;                              // input cells per side of a tile's neighbourhood
constexpr int kFInPieces = kFIn * (kFIn * 8 / 16);    // 968 16-byte pieces
constexpr int kFInDma = (kFInPieces + 63) / 64;       // 16 wave-wide copies (the last one's surplus lands in patch buffer 0)
constexpr int kFInBytes = kFIn * kFIn * 8;            // 15 488
constexpr int kFPatchElems = kG2Pix * 16;             // one channel group of a patch, in 4-byte elements
constexpr int kFLds = kFInBytes + 2 * kFPatchElems * 4;

// produce(G): rows of the patch are wave tiles of 16 pool windows (14 used).  `cbase[j]` is the cell
// (within an input row) that A-row lane&15 of column group j reads, kernel column lane>>4 included.
template <int G>
__device__ __forceinline__ void conv12_produce(const uint2* cin, uint32_t* pb, const u32x4 (&bw)[3][3],
                                               const float (&p1)[3][3], const int (&cbase)[3], int lane,
                                               int Y0, int X0, int H1) {
    const int q = lane >> 4, ch = lane & 15;
    const float bias = p1[G][0], sc = p1[G][1], sh = p1[G][2];
    const f32x4 bias4 = f32x4{bias, bias, bias, bias};
    uint16_t* out = reinterpret_cast<uint16_t*>(pb) + (ch >> 2) * 8 + (ch & 3);     // + pixel * 32; lo at + 4
    auto cell = [&](int row, int j) {
        const uint2 c = cin[cbase[j] + row * kFIn];
        return u32x4{c.x, c.y, c.x, c.y};
    };
    auto finish = [&](const f32x4 (&acc)[3][3], int r) {
        const float m[4] = {window_max<0>(acc), window_max<1>(acc), window_max<2>(acc), window_max<3>(acc)};
        // rows and columns outside the frame are block 2's zero padding: a zero scale and shift for such a
        // row (selects, so that the row's code stays one basic block the scheduler can interleave), a
        // select per column
        const bool row_in = (unsigned)(Y0 + r) < (unsigned)H1;
        const float scr = row_in ? sc : 0.f, shr = row_in ? sh : 0.f;
        float v[4];
#pragma unroll
        for (int wi = 0; wi < 4; ++wi) {
            const bool in = (unsigned)(X0 + 4 * q + wi) < (unsigned)kW1;
            v[wi] = in ? fmaf(m[wi], scr, shr) : 0.f;
        }
        // windows 14 and 15 of the row (lanes q = 3, wi = 2, 3) are not part of the patch: they write
        // window 13's value to window 13's cell again instead of branching around the stores
        const bool tail = q == 3;
        v[2] = tail ? v[1] : v[2];
        v[3] = tail ? v[1] : v[3];
#pragma unroll
        for (int wi = 0; wi < 4; ++wi) {
            const int x = wi < 2 ? 4 * q + wi : (tail ? 13 : 4 * q + wi);
            const uint32_t hi = pack_bf16(v[wi], 0.f) & 0xFFFFu;
            const uint32_t lo = pack_bf16(v[wi] - __uint_as_float(hi << 16), 0.f);
            out[(r * kP2 + x) * 32] = (uint16_t)hi;
            out[(r * kP2 + x) * 32 + 4] = (uint16_t)lo;
        }
    };
    auto mfmas = [&](const u32x4 (&a)[5][3], f32x4 (&acc)[3][3]) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[t + ky][j]),
                                                                        __builtin_bit_cast(bf16x8, bw[ky][G]),
                                                                        ky == 0 ? bias4 : acc[t][j], 0, 0, 0);
    };
    // Scheduling pattern of one patch row: its 27 MFMAs with, behind each, a share of the other row's
    // pooling / affine / split arithmetic and of the LDS traffic (the next row's cell reads early, the
    // other row's result writes late).  Left to itself the compiler issues the MFMAs back to back and
    // the vector work after them, and the wave does one thing at a time.
    auto interleave = [&]() {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
            if (k < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                 // four vector instructions
            if (k >= 16 && k < 24) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // one LDS write
        }
    };
    u32x4 a[5][3];
#pragma unroll
    for (int rr = 0; rr < 5; ++rr)
#pragma unroll
        for (int j = 0; j < 3; ++j) a[rr][j] = cell(rr, j);
    // Two patch rows per iteration, each with accumulators of its own: a row's pooling / affine / split
    // / LDS writes are independent of the other row's MFMAs and are scheduled among them.
    f32x4 acc0[3][3], acc1[3][3];
    auto advance = [&](int r) {          // a <- the five input rows of patch row r + 1
        u32x4 nx[3][3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int j = 0; j < 3; ++j) nx[rr][j] = cell(3 * (r + 1) + 2 + rr, j);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a[0][j] = a[3][j]; a[1][j] = a[4][j];
            a[2][j] = nx[0][j]; a[3][j] = nx[1][j]; a[4][j] = nx[2][j];
        }
    };
#pragma unroll
    for (int r = 0; r < kP2; r += 2) {
        // (sched_barrier: scheduling stays within a row; left free across rows, the compiler requests
        // every row's cells at once and spills)
        mfmas(a, acc0);
        if (r > 0) finish(acc1, r - 1);
        advance(r);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a, acc1);
        finish(acc0, r);
        if (r + 2 < kP2) advance(r + 1);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    }
    finish(acc1, kP2 - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// byte offset of lane's piece of wave-wide copy `it` of a tile's input cells (22 pieces per 44-cell row)
__device__ __forceinline__ int conv12_in_rel(int it, int lane) {
    const int f = it * 64 + lane, row = min(f / 22, kFIn - 1), pc = f % 22;
    return row * (kCellPitch * 8) + pc * 16;
}

// LDS-DMA of a tile's 44 x 44 input cells: 15 full wave-wide copies and one of 8 lanes
__device__ __forceinline__ void conv12_copy_cells(const char* src, char* cin_lds, const int (&rel)[kFInDma], int lane) {
#pragma unroll
    for (int it = 0; it < kFInDma; ++it) {
        if (it < kFInDma - 1 || lane < kFInPieces - 64 * (kFInDma - 1))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint32_t)rel[it]),
                                             (__attribute__((address_space(3))) void*)(cin_lds + it * 1024), 16, 0, 0);
    }
}

// consume(G): the 9 steps x 54 MFMAs of one channel group of a block-2 wave tile on a patch that is
// complete in LDS.  The operands of the next step (nine LDS reads, three weight fragments through the
// buffer descriptor) are requested one per MFMA during the current step; the A operands of a unit's
// first step are read at its start (the patch did not exist earlier), its weights were requested
// during the previous unit's last step.
// Byte offset, from a patch's first activation, of the 16 bytes lane `lane` copies in copy instruction
// `it` of a 14x14 split-bf16 patch (4 pieces per pixel; same values as conv2_kernel's rel[] table, worked
// out from two compares instead of held in 13 registers).  rel_lane = (lane / 4) * 192 + (lane % 4) * 16.
__device__ __forceinline__ int conv2_rel(int it, int lane, int rel_lane) {
    const int start = 16 * it, row0 = start / kP2, col0 = start % kP2;    // constants once unrolled
    const int q = lane >> 2;
    const int wraps = (q >= kP2 - col0 ? 1 : 0) + (q >= 2 * kP2 - col0 ? 1 : 0);
    return (row0 * kP1W + col0) * kC * 4 + rel_lane + wraps * ((kP1W - kP2) * kC * 4);
}

template <int G>
__device__ __forceinline__ void conv12_consume(const uint32_t* pb, const int (&base)[3], __amdgpu_buffer_rsrc_t wr,
                                               int lane, u32x4 (&a)[2][3][3], SplitB (&b)[2][3], f32x4 (&acc)[3][3][3]) {
    constexpr int kM = 54, kFirst = G * 9;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr int c0 = kFirst & 1;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) a[c0][t][j] = *reinterpret_cast<const u32x4*>(pb + base[j] + t * kP2 * 16);
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) {
        const int c = (kFirst + kk) & 1;
        const int bstep = (kFirst + kk + 1) % 27, bg = bstep / 9, bk = bstep % 9;
        const int nk = kk + 1;
        const int off = ((nk / 3) * kP2 + (nk % 3)) * 16;
        const int n_side = kk < 8 ? 12 : 3;
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            const int e = m / 27, t = (m / 9) % 3, j = (m / 3) % 3, nt = m % 3;
            if (G == 0 && kk == 0 && e == 0)
                acc[t][j][nt] = mfma_frag<kS2>(a[c][t][j], b[c][nt], e, f32x4{0.f, 0.f, 0.f, 0.f});
            else
                acc[t][j][nt] = mfma_frag<kS2>(a[c][t][j], b[c][nt], e, acc[t][j][nt]);
            if (m < n_side) {
                __builtin_amdgcn_sched_barrier(0);
                if (m < 3) {
                    b[c ^ 1][m] = load_frag<kS2>(wr, lane, (bk * 3 + bg) * 3 + m);
                } else {
                    const int lt = (m - 3) / 3, lj = (m - 3) % 3;
                    a[c ^ 1][lt][lj] = *reinterpret_cast<const u32x4*>(pb + base[lj] + off + lt * kP2 * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Workgroup = two waves with two jobs: wave 0 produces the patches (block 1), wave 1 consumes them
// (block 2).  Step S of a workgroup: the producer writes channel group S % 3 of tile S / 3 into patch
// buffer S & 1 while the consumer works on the group written in step S - 1; a workgroup barrier ends
// the step.  Two waves per SIMD (256 registers each): one wave's pooling arithmetic, LDS traffic and
// waits run in the shadow of the other's MFMAs.
__global__ __launch_bounds__(128, 2) void conv12_split_kernel(const uint8_t* __restrict__ cells, uint32_t* __restrict__ out,
                                                              const u32x4* __restrict__ w1, const float* __restrict__ post1,
                                                              const void* __restrict__ w2, const float* __restrict__ post2,
                                                              int n, int H, int H1, int Hp, int tiles_x, int tiles_y,
                                                              int total_tiles, unsigned* __restrict__ tickets) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kFLds];
    uint2* cin = reinterpret_cast<uint2*>(lds);
    uint32_t* pb0 = reinterpret_cast<uint32_t*>(lds + kFInBytes);
    const int lane = threadIdx.x & 63;
    // Which wave produces.  The dispatcher deals the waves of a CU's workgroups round the SIMDs in a
    // fixed order, so with four of these workgroups on a CU the first waves of two of them share one
    // SIMD and their second waves another: "wave 0 produces" would put two producers on one SIMD and
    // two consumers on the other.  Every workgroup therefore takes a ticket from a counter of its
    // (CU, SIMD of wave 0): the parity of the ticket swaps the roles in every second workgroup there,
    // and each SIMD gets one producer (vector arithmetic + short MFMA bursts) beside one consumer
    // (MFMAs back to back).  Only the balance depends on this; any assignment is correct.
    __shared__ int flip_lds;
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned slot = ((xcc & 7u) << 8) | (((hw >> 13) & 3u) << 6) | (((hw >> 8) & 15u) << 2) | ((hw >> 4) & 3u);
        flip_lds = (int)(atomicAdd(tickets + slot, 1u) & 1u);
    }
    __syncthreads();
    const bool producer = (threadIdx.x < 64) != (flip_lds != 0);
    const int per_frame = tiles_x * tiles_y;
    const int64_t in_frame = (int64_t)(H + 2 * kCellPad) * kCellPitch * 8, out_frame = (int64_t)(Hp + 2) * kP2W * kC;
    auto valid = [&](int id) {
        int frame, tt;
        xcd_frame_tile(id, per_frame, frame, tt);
        return id >= 0 && id < total_tiles && frame < n;
    };
    struct Tile { const char* src; uint32_t* dst; int ty, tx; };
    auto locate = [&](int id) {
        int frame = 0, tt = 0;
        xcd_frame_tile(id, per_frame, frame, tt);
        Tile t;
        t.ty = tt / tiles_x; t.tx = tt % tiles_x;
        t.src = reinterpret_cast<const char*>(cells) + frame * in_frame + ((int64_t)(36 * t.ty) * kCellPitch + 36 * t.tx) * 8;
        t.dst = out + frame * out_frame;
        return t;
    };
    // Both waves walk the same tiles (ids blockIdx.x, + gridDim.x, ...: validity only ever goes from
    // true to false along that sequence, frames being dealt to ids in increasing order) and execute
    // the same number of barriers: one per step, plus the consumer's lead-in / the producer's lead-out.
    int id = blockIdx.x;
    if (!valid(id)) return;
    const int i = lane & 15, kq = lane >> 4;
#ifdef CD_DIAG12
    // diagnostic build: per-wave time stamps of the second and third tile go to the slack frame behind the batch
    unsigned long long* st12 = reinterpret_cast<unsigned long long*>(out + (int64_t)n * out_frame) + (blockIdx.x * 2 + (producer ? 0 : 1)) * 24;
    int tile_no = 0;
#define CD12_STAMP(k) do { if (lane == 0 && blockIdx.x < 256 && tile_no >= 1 && tile_no <= 2) st12[(tile_no - 1) * 12 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CD12_NEXT() (++tile_no)
#else
#define CD12_STAMP(k) do { } while (0)
#define CD12_NEXT() do { } while (0)
#endif

    if (producer) {
        int cbase[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c12 = 4 * j + (i & 3), d = (c12 * 11) >> 5, m = c12 - 3 * d;      // d = c12 / 3
            const int W = min(4 * (i >> 2) + d, kP2 - 1);
            // (lanes kq = 3 carry zero weights; they re-read the kq = 2 cell)
            cbase[j] = 3 * W + m + min(kq, 2);
        }
        // block-1 weight fragments (ky, n-tile) and its bias / BatchNorm terms stay in registers
        u32x4 bw[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) bw[ky][nt] = w1[(ky * 3 + nt) * 64 + lane];
        float p1[3][3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int k = 0; k < 3; ++k) p1[nt][k] = post1[k * kC + nt * 16 + (lane & 15)];
        int rel[kFInDma];
#pragma unroll
        for (int it = 0; it < kFInDma; ++it) rel[it] = conv12_in_rel(it, lane);
        Tile cur = locate(id);
        conv12_copy_cells(cur.src, reinterpret_cast<char*>(lds), rel, lane);
        int par = 0;                      // patch buffer of the next group
        for (;;) {
            CD12_STAMP(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this tile's input cells have landed
            CD12_STAMP(1);
            const int Y0 = 12 * cur.ty - 1, X0 = 12 * cur.tx - 1;
            conv12_produce<0>(cin, pb0 + par * kFPatchElems, bw, p1, cbase, lane, Y0, X0, H1);
            CD12_STAMP(2);
            __builtin_amdgcn_s_barrier();
            CD12_STAMP(3);
            par ^= 1;
            conv12_produce<1>(cin, pb0 + par * kFPatchElems, bw, p1, cbase, lane, Y0, X0, H1);
            CD12_STAMP(4);
            __builtin_amdgcn_s_barrier();
            CD12_STAMP(5);
            par ^= 1;
            conv12_produce<2>(cin, pb0 + par * kFPatchElems, bw, p1, cbase, lane, Y0, X0, H1);
            CD12_STAMP(6);
            par ^= 1;
            // every read of the input cells is done (produce ends with an LDS wait): the next tile's
            // cells may replace them
            const int next_id = id + (int)gridDim.x;
            const bool more = valid(next_id);
            if (more) {
                id = next_id;
                cur = locate(id);
                conv12_copy_cells(cur.src, reinterpret_cast<char*>(lds), rel, lane);
            }
            CD12_STAMP(7);
            __builtin_amdgcn_s_barrier();
            CD12_STAMP(8);
            CD12_NEXT();
            if (!more) break;
        }
        __builtin_amdgcn_s_barrier();          // the consumer's last step
        return;
    }

    // ---- consumer
    int base[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) base[j] = ((3 * (i >> 2)) * kP2 + 4 * j + (i & 3)) * 16 + 4 * kq;
    const __amdgpu_buffer_rsrc_t wr = weight_rsrc<kS2>(w2);
    float p2[3][3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int k = 0; k < 3; ++k) p2[nt][k] = post2[k * kC + nt * 16 + (lane & 15)];
    u32x4 a[2][3][3];
    SplitB b[2][3];
    f32x4 acc[3][3][3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) b[0][nt] = load_frag<kS2>(wr, lane, nt);      // step 0: tap 0 of channel group 0
    int par = 0;
    __builtin_amdgcn_s_barrier();              // the producer's first step
    for (;;) {
        const Tile cur = locate(id);
        CD12_STAMP(0);
        conv12_consume<0>(pb0 + par * kFPatchElems, base, wr, lane, a, b, acc);
        CD12_STAMP(1);
        __builtin_amdgcn_s_barrier();
        CD12_STAMP(2);
        par ^= 1;
        conv12_consume<1>(pb0 + par * kFPatchElems, base, wr, lane, a, b, acc);
        CD12_STAMP(3);
        __builtin_amdgcn_s_barrier();
        CD12_STAMP(4);
        par ^= 1;
        conv12_consume<2>(pb0 + par * kFPatchElems, base, wr, lane, a, b, acc);
        CD12_STAMP(5);
        // pooled + normalised results -> HBM, transposed through the patch buffer just consumed (the
        // producer is writing the other one; it gets this one back at the barrier below)
        {
            float* scratch = reinterpret_cast<float*>(pb0 + par * kFPatchElems);
            const int q = lane >> 4, ch = lane & 15;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int c = nt * 16 + ch;
                const float m[4] = {window_max<0>(acc, nt), window_max<1>(acc, nt), window_max<2>(acc, nt),
                                    window_max<3>(acc, nt)};
#pragma unroll
                for (int wi = 0; wi < 4; ++wi)
                    scratch[(4 * q + wi) * kC + c] = fmaf(fmaxf(m[wi] + p2[nt][0], 0.f), p2[nt][1], p2[nt][2]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int f = it * 64 + lane;                 // 16-byte piece f of 16 windows x 192 bytes
                const int win = f / 12, c4 = f % 12;
                const int py = cur.ty * 4 + (win >> 2), px = cur.tx * 4 + (win & 3);
                const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + win * kC + 4 * c4);
                if (py < Hp && px < kW2)
                    *reinterpret_cast<u32x4*>(cur.dst + ((int64_t)(py + 1) * kP2W + px + 1) * kC + 4 * c4) = split_bf16(v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        par ^= 1;
        // 27 steps leave the next step's weights (tap 0 of group 0 again) in set 1; a tile starts from set 0
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) b[0][nt] = b[1][nt];
        const int next_id = id + (int)gridDim.x;
        const bool more = valid(next_id);
        CD12_STAMP(6);
        __builtin_amdgcn_s_barrier();
        CD12_STAMP(7);
        CD12_NEXT();
        if (!more) break;
        id = next_id;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// Block 2 in split bf16 with TWO waves per SIMD.
// With 16-cycle MFMAs the issue cost of a step's memory instructions no longer hides in the MFMA gaps:
// alone on its SIMD, conv2_kernel<kS2>'s wave keeps the matrix pipe ~65 % busy.  Here a second wave
// fills the gaps.  To fit two, a wave has 256 registers and ONE patch buffer:
//   - a step's MFMAs run patch row by patch row (t outermost), so only the first row's operands have to
//     be there when the step starts: they are read during the previous step (12 registers), the other
//     two rows at the start of the step itself, behind its first MFMAs (no second 36-register set);
//   - the weights (16 bytes per lane and fragment, {hi x 4, lo x 4}) are requested kS2Ahead steps
//     ahead: they come from L2 (a CU's L1 does not hold the 83 KB a tile cycles through), behind the
//     other waves' patch copies, and one step of a wave that shares its SIMD is not always enough;
//   - the next unit's patch is copied over the current one during the unit's last step, once that
//     step's own operands have been read; what is left of the copy's latency is the other wave's time
//     on the pipe.
// Every accumulator sees its products in the order conv2_kernel<kS2> adds them: same bits.
// ------------------------------------------------------------------------------------------
#ifndef CD_S2W2_AHEAD
#define CD_S2W2_AHEAD 2
#endif
constexpr int kS2Ahead = CD_S2W2_AHEAD, kS2Sets = kS2Ahead + 1;
constexpr int kS2W2WaveLds = 16384;      // per wave: one patch buffer (13 312 bytes) + the epilogue's 16 x 48 floats
static_assert(27 % kS2Sets == 0, "the weight register sets must come round at a tile boundary");

template <int G, int WAVES>
__device__ __forceinline__ void conv2_split2_unit(uint32_t* pb, const int (&base)[3], __amdgpu_buffer_rsrc_t wr, int lane,
                                                  u32x4 (&a)[3][3], u32x4 (&an)[3], SplitB (&b)[kS2Sets][3],
                                                  f32x4 (&acc)[3][3][3], const char* next_src, int rel_lane) {
    constexpr int kDma = G2<kS2>::kDma;
    // the waves of a workgroup start a unit together: they then ask for the same weight lines within a few
    // hundred cycles of each other, and all but the first request are served by the CU's L1
    if (WAVES > 1) __builtin_amdgcn_s_barrier();
    // the patch (copied during the previous unit's last step, or by the prologue) must have landed; the
    // three weight loads issued after the copy may still be in flight (VMEM completes in issue order)
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) a[0][j] = *reinterpret_cast<const u32x4*>(pb + base[j]);
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) {
        constexpr int kFirst = G * 9;
        const int s = kFirst + kk, cb = s % kS2Sets, lb = (s + kS2Ahead) % kS2Sets;
        const int sw = (s + kS2Ahead) % 27, wk = sw % 9, wg = sw / 9;          // the step whose weights are requested now
        const int off = (kk / 3) * kP2 * 16 + (kk % 3) * 16;
        const int noff = ((kk + 1) / 3) * kP2 * 16 + ((kk + 1) % 3) * 16;
#pragma unroll
        for (int m = 0; m < 54; ++m) {
            const int t = m / 18, e = (m / 9) % 2, j = (m / 3) % 3, nt = m % 3;
            if (G == 0 && kk == 0 && e == 0)
                acc[t][j][nt] = mfma_frag<kS2>(a[t][j], b[cb][nt], e, f32x4{0.f, 0.f, 0.f, 0.f});
            else
                acc[t][j][nt] = mfma_frag<kS2>(a[t][j], b[cb][nt], e, acc[t][j][nt]);
            const int n_side = kk < 8 ? 12 : 9 + kDma;
            if (m < n_side) {
                __builtin_amdgcn_sched_barrier(0);
                if (m < 6) {                        // this step's rows 1 and 2
                    const int lt = 1 + m / 3, lj = m % 3;
                    a[lt][lj] = *reinterpret_cast<const u32x4*>(pb + base[lj] + off + lt * kP2 * 16);
                } else if (kk < 8) {
                    if (m < 9) b[lb][m - 6] = load_frag<kS2>(wr, lane, (wk * 3 + wg) * 3 + (m - 6));
                    else an[m - 9] = *reinterpret_cast<const u32x4*>(pb + base[m - 9] + noff);      // next step's row 0
                } else {
                    // last step: nothing reads the patch after the six reads above
                    if (m == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (m < 6 + kDma) {
                        const int it = m - 6;
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(next_src + (uint32_t)conv2_rel(it, lane, rel_lane)),
                            (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(pb) + it * 1024), 16, 0, CD_DMA_AUX);
                    } else {
                        b[lb][m - 6 - kDma] = load_frag<kS2>(wr, lane, (wk * 3 + wg) * 3 + (m - 6 - kDma));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kk < 8) {
#pragma unroll
            for (int j = 0; j < 3; ++j) a[0][j] = an[j];
        }
    }
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 1 ? 2 : 1) void conv2_split2_kernel(
    const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const void* __restrict__ wfrag,
    const float* __restrict__ post, int n, int Hin, int Hp, int tiles_x, int tiles_y, int total_tiles) {
    constexpr int kDma = G2<kS2>::kDma, kBuf = G2<kS2>::kBufElems;
    static_assert((kBuf + 16 * kC) * 4 == kS2W2WaveLds, "LDS per wave");
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_w2[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* patch = dyn_w2 + wave * (kS2W2WaveLds / 4);
    float* scratch = reinterpret_cast<float*>(patch + kBuf);             // the epilogue's transpose
    const int slot = (int)blockIdx.x * WAVES + wave, stride = (int)gridDim.x * WAVES;
    const int per_frame = tiles_x * tiles_y;
    const int64_t in_frame = (int64_t)(Hin + 2) * kP1W * kC * 4, out_frame = (int64_t)(Hp + 2) * kP2W * kC;
    auto valid = [&](int id) {
        int frame, tt;
        xcd_frame_tile(id, per_frame, frame, tt);
        return id >= 0 && id < total_tiles && frame < n;
    };
    // invalid ids map to tile 0 of frame 0: the copy a wave's last unit issues stays in bounds and is never used
    auto locate = [&](int id) {
        int frame = 0, tt = 0;
        if (valid(id)) xcd_frame_tile(id, per_frame, frame, tt);
        Conv2Tile<kS2> t;
        t.ty = tt / tiles_x; t.tx = tt % tiles_x;
        t.src = reinterpret_cast<const char*>(in) + frame * in_frame + ((int64_t)(12 * t.ty) * kP1W + 12 * t.tx) * kC * 4;
#ifdef CD_EXP_DMAHOT
        t.src = reinterpret_cast<const char*>(in);       // timing experiment: every patch copy hits the caches (wrong results)
#endif
        t.dst = out + frame * out_frame;
        return t;
    };
    int id = slot;
    // every wave of a workgroup meets the others at three barriers per round, tiles or not
    int rounds_left = WAVES > 1 ? (total_tiles + stride - 1) / stride : 0;
    if (!valid(id)) {
        for (; rounds_left > 0; --rounds_left)
            for (int g = 0; g < 3; ++g) __builtin_amdgcn_s_barrier();
        return;
    }
    const int rel_lane = (lane >> 2) * kC * 4 + (lane & 3) * 16;
    const int i = lane & 15, kq = lane >> 4;
    int base[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) base[j] = ((3 * (i >> 2)) * kP2 + 4 * j + (i & 3)) * 16 + 4 * kq;
    const __amdgpu_buffer_rsrc_t wr = weight_rsrc<kS2>(wfrag);
    Conv2Tile<kS2> cur = locate(id);
    int next_id = valid(id + stride) ? id + stride : -1;
    Conv2Tile<kS2> nxt = locate(next_id);
    u32x4 a[3][3], an[3];
    SplitB b[kS2Sets][3];
    f32x4 acc[3][3][3];
#pragma unroll
    for (int it = 0; it < kDma; ++it)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(cur.src + (uint32_t)conv2_rel(it, lane, rel_lane)),
            (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(patch) + it * 1024), 16, 0, CD_DMA_AUX);
    // the weights of the first kS2Ahead steps (step s of a tile = kernel tap s % 9 of channel group s / 9)
#pragma unroll
    for (int s0 = 0; s0 < kS2Ahead; ++s0)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) b[s0][nt] = load_frag<kS2>(wr, lane, ((s0 % 9) * 3 + s0 / 9) * 3 + nt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr int kGroupBytes = 64;
    for (;;) {
        conv2_split2_unit<0, WAVES>(patch, base, wr, lane, a, an, b, acc, cur.src + kGroupBytes, rel_lane);
        conv2_split2_unit<1, WAVES>(patch, base, wr, lane, a, an, b, acc, cur.src + 2 * kGroupBytes, rel_lane);
        conv2_split2_unit<2, WAVES>(patch, base, wr, lane, a, an, b, acc, nxt.src, rel_lane);
        --rounds_left;
        {
            // bias / BatchNorm constants are fetched here (L1 hits) rather than held in registers all tile long
            const int q = lane >> 4, ch = lane & 15;
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int c = nt * 16 + ch;
                const float bias = post[c], scale = post[kC + c], shift = post[2 * kC + c];
                const float m[4] = {window_max<0>(acc, nt), window_max<1>(acc, nt), window_max<2>(acc, nt),
                                    window_max<3>(acc, nt)};
#pragma unroll
                for (int wi = 0; wi < 4; ++wi)
                    scratch[(4 * q + wi) * kC + c] = fmaf(fmaxf(m[wi] + bias, 0.f), scale, shift);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int f = it * 64 + lane;                 // 16-byte piece f of 16 windows x 192 bytes
                const int win = f / 12, c4 = f % 12;
                const int py = cur.ty * 4 + (win >> 2), px = cur.tx * 4 + (win & 3);
                const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + win * kC + 4 * c4);
                if (py < Hp && px < kW2)
                    *reinterpret_cast<u32x4*>(cur.dst + ((int64_t)(py + 1) * kP2W + px + 1) * kC + 4 * c4) = split_bf16(v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (next_id < 0) break;
        id = next_id;
        cur = nxt;
        next_id = valid(id + stride) ? id + stride : -1;
        nxt = locate(next_id);
    }
    // the copy and the weight loads of a unit that will never run are still in flight
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (; rounds_left > 0; --rounds_left)
        for (int g = 0; g < 3; ++g) __builtin_amdgcn_s_barrier();
}

#ifndef CD_S2W2_WAVES
#define CD_S2W2_WAVES 1
#endif
int init_conv2_split2() {
    CD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_split2_kernel<CD_S2W2_WAVES>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, CD_S2W2_WAVES * kS2W2WaveLds));
    return CD_OK;
}

int launch_conv2_split2(const cd_handle* h, const Workspace& ws, int n, const Geometry& g, hipStream_t s) {
    constexpr int kWaves = CD_S2W2_WAVES;
    const int tiles_y = (g.h2 + 3) / 4, tiles_x = (kW2 + 3) / 4;
    const int total = (int)xcd_grid(n, tiles_x * tiles_y);
    const int want = (total + kWaves - 1) / kWaves, grid = want < 2048 / kWaves ? want : 2048 / kWaves;
    hipLaunchKernelGGL(conv2_split2_kernel<kWaves>, dim3((unsigned)grid), dim3(64 * kWaves), kWaves * kS2W2WaveLds, s,
                       reinterpret_cast<const uint32_t*>(ws.act1), reinterpret_cast<uint32_t*>(ws.act2),
                       (const void*)h->p.w2s, h->p.post + 3 * kC, n, g.h1, g.h2, tiles_x, tiles_y, total);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

int launch_conv12_split(const cd_handle* h, const Workspace& ws, const uint8_t* cells, int n, const Geometry& g,
                        hipStream_t s) {
    const int tiles_y = (g.h2 + 3) / 4, tiles_x = (kW2 + 3) / 4;
    const int total = (int)xcd_grid(n, tiles_x * tiles_y);
    const int grid = total < 1024 ? total : 1024;
    hipLaunchKernelGGL(conv12_split_kernel, dim3((unsigned)grid), dim3(128), 0, s, cells, reinterpret_cast<uint32_t*>(ws.act2),
                       reinterpret_cast<const u32x4*>(h->p.w1s), (const float*)h->p.post, (const void*)h->p.w2s,
                       h->p.post + 3 * kC, n, g.h, g.h1, g.h2, tiles_x, tiles_y, total, h->tickets);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

// ------------------------------------------------------------------------------------------
// Block 3 + head: one workgroup (4 waves) per frame.
// The whole zero-padded block-2 activation ((Hin+2) x 30 x 48 fp32) sits in LDS; pool windows
// are enumerated row-major, 16 per wave tile (45 windows = 3 tiles at 144 input rows).  The
// pooled, normalised 5x9x48 tensor goes to LDS, then adaptive average pooling (bins
// [floor(i*H/4), ceil((i+1)*H/4)) as torch defines them, net.py:88,130), the channel-major
// flatten (net.py:131), Linear(768,32)+ReLU+BatchNorm1d, Linear(32,3) (net.py:62-68) and the
// max/argmax of segmentation.py:37 finish the frame.
// ------------------------------------------------------------------------------------------
constexpr int kP3W = kW2 + 2;   // 30

template <int BF>
__global__ __launch_bounds__(256) void conv3_head_kernel(
    const typename Prec<BF>::elem* __restrict__ in, const void* __restrict__ wfrag, const float* __restrict__ post,
    const float* __restrict__ fc0w, const float* __restrict__ fc0p, const float* __restrict__ fc1w,
    const float* __restrict__ fc1b, float* __restrict__ logits, float* __restrict__ maxv,
    int64_t* __restrict__ argmax, int Hin, int Hp, int spare_off) {
    typedef typename Prec<BF>::elem elem;
    constexpr int kEs = Prec<BF>::kEs;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    elem* patch = reinterpret_cast<elem*>(smem);          // [(Hin+2)][30][48]
    float* pooled = smem + (Hin + 2) * kP3W * kC * kEs / 4;   // [Hp][9][48]
    float* feat = pooled + Hp * kW3 * kC;                 // [768]
    float* hid = feat + kFeat;                            // [32]
    float* spost = hid + kHid;                            // [3][48] bias, BN scale, BN shift of block 3
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int q = lane >> 4, ch = lane & 15;

#ifdef CD_DIAG3   // (shares the slack frame with block 2's stamps: a build of its own, -DCD_DIAG -DCD_DIAG3)
    // diagnostic build: per-wave time stamps go to the slack frame behind the batch's activations
    unsigned long long* st3 = reinterpret_cast<unsigned long long*>(const_cast<elem*>(in) + (int64_t)gridDim.x * (Hin + 2) * kP3W * kC) + (blockIdx.x * 4 + wave) * 8;
    int nst3 = 0;
#define CD_STAMP3() do { if (BF == CD_DIAG_P && lane == 0 && blockIdx.x < 400) st3[nst3] = __builtin_amdgcn_s_memrealtime(); ++nst3; } while (0)
#else
#define CD_STAMP3() do { } while (0)
#endif
    CD_STAMP3();
    // the padded block-2 activation of this frame is contiguous in HBM: a linear LDS-DMA copy
    const char* fin = reinterpret_cast<const char*>(in + (int64_t)frame * (Hin + 2) * kP3W * kC);
    const int patch_bytes = (Hin + 2) * kP3W * kC * kEs;
    for (int off0 = (tid >> 6) * 1024; off0 < patch_bytes; off0 += 4096) {
        const int off = off0 + (tid & 63) * 16;
        if (off < patch_bytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(fin + off),
                                             (__attribute__((address_space(3))) void*)((char*)patch + off0), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CD_STAMP3();
    // The small parameters that do not depend on the activations are requested now and land while
    // the convolution runs: a dependent global load costs more than a microsecond here, and the head
    // would otherwise pay that once per layer.
    const int fo = tid >> 3, part = tid & 7;
    f32x4 w0[24];
    // 16-byte piece k*8 + part of the output's weight row: the 8 threads of an output read 128
    // contiguous bytes per load.  Requested when the convolution and its epilogue are over (their
    // accumulators need the registers) and used two barriers later.
    auto load_fc0 = [&]() {
        const f32x4* wrow = reinterpret_cast<const f32x4*>(fc0w + fo * kFeat) + part;
#pragma unroll
        for (int k = 0; k < 24; ++k) w0[k] = wrow[k * 8];
    };
    const float h_bias = fc0p[fo], h_scale = fc0p[kHid + fo], h_shift = fc0p[2 * kHid + fo];
    if (tid < 3 * kC) spost[tid] = post[tid];     // visible to the epilogues after the barriers that follow
    const float l_w01 = fc1w[min(tid >> 5, 1) * kHid + (tid & 31)], l_w2 = fc1w[2 * kHid + (tid & 31)];
    const float l_b0 = fc1b[0], l_b1 = fc1b[1], l_b2 = fc1b[2];

    const int windows = Hp * kW3;
    const int wave_tiles = (windows + 15) / 16;
    const __amdgpu_buffer_rsrc_t wr = weight_rsrc<BF>(wfrag);
    // LDS offsets of the pixels that A-row i of column group j maps to in wave tile wt.  Columns of
    // one lane's 12 may straddle pool rows here, so tile (t, j) of lane-row i uses base[j] for every
    // t and the conv row offset t is applied inside the tile routine.
    auto tile_base = [&](int wt, int (&base)[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int c12 = 4 * j + (i & 3);
            int W = min(wt * 16 + 4 * (i >> 2) + c12 / 3, windows - 1);
            int wrow = W / kW3, px = W % kW3;
            base[j] = ((3 * wrow) * kP3W + 3 * px + c12 % 3) * kC + 4 * kq;
        }
    };
    // ReLU + 3x3 max + bias + BatchNorm affine of one wave tile -> pooled[]
    auto tile_epilogue = [&](int wt, const f32x4 (&acc)[3][3][3]) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int c = nt * 16 + ch;
            const float bias = spost[c], sc = spost[kC + c], sh = spost[2 * kC + c];
            float m[4] = {window_max<0>(acc, nt), window_max<1>(acc, nt), window_max<2>(acc, nt),
                          window_max<3>(acc, nt)};
#pragma unroll
            for (int wi = 0; wi < 4; ++wi) {
                int W = wt * 16 + 4 * q + wi;
                if (W < windows) pooled[W * kC + c] = fmaf(fmaxf(m[wi] + bias, 0.f), sc, sh);
            }
        }
    };
    // Up to three wave tiles (45 windows at 144 input rows) would leave the fourth wave, a quarter of
    // the CU's matrix throughput, idle.  Then (the host passes spare_off >= 0) the K range is split
    // instead: waves 0-2 run steps 0..20 of their tile, wave 3 runs steps 21..26 of all of them (18
    // steps, plus the hand-over of its partial sums through LDS; 21 measured best of 19..23).
    constexpr int kSplit = 21;
    if (spare_off >= 0) {
        if constexpr (BF != kBF16) {
            // wave 3 keeps at most two tiles' partial sums in registers: the first tile's go to a
            // spare LDS buffer behind the head's arrays as soon as they are complete
            f32x4* spare = reinterpret_cast<f32x4*>(smem + spare_off);   // [27 accumulators][64 lanes]
            f32x4* parts = reinterpret_cast<f32x4*>(patch);              // same, per tile, once the patch is dead
            f32x4 acc[3][3][3], extra[3][3][3];
            int base[3];
            if (wave < 3) {
                if (wave < wave_tiles) {
                    tile_base(wave, base);
                    conv48_wave_tile<kP3W, BF, 0, kSplit>(patch, base, wr, lane, acc);
                }
            } else {
                tile_base(0, base);
                conv48_wave_tile<kP3W, BF, kSplit, 27>(patch, base, wr, lane, acc);
#pragma unroll
                for (int k = 0; k < 27; ++k) spare[k * 64 + lane] = acc[k / 9][(k / 3) % 3][k % 3];
                if (wave_tiles > 1) { tile_base(1, base); conv48_wave_tile<kP3W, BF, kSplit, 27>(patch, base, wr, lane, acc); }
                if (wave_tiles > 2) { tile_base(2, base); conv48_wave_tile<kP3W, BF, kSplit, 27>(patch, base, wr, lane, extra); }
            }
            CD_STAMP3();
            __syncthreads();                                   // every wave is done reading the patch
            if (wave == 3) {
#pragma unroll
                for (int k = 0; k < 27; ++k) {
                    if (wave_tiles > 1) parts[(27 + k) * 64 + lane] = acc[k / 9][(k / 3) % 3][k % 3];
                    if (wave_tiles > 2) parts[(54 + k) * 64 + lane] = extra[k / 9][(k / 3) % 3][k % 3];
                }
            }
            __syncthreads();
            if (wave < 3 && wave < wave_tiles) {
                const f32x4* mine = wave == 0 ? spare : parts + wave * 27 * 64;
#pragma unroll
                for (int k = 0; k < 27; ++k) acc[k / 9][(k / 3) % 3][k % 3] += mine[k * 64 + lane];
                tile_epilogue(wave, acc);
            }
            load_fc0();
            CD_STAMP3();
        }
    } else {
        for (int wt = wave; wt < wave_tiles; wt += 4) {
            int base[3];
            tile_base(wt, base);
            f32x4 acc[3][3][3];
            conv48_wave_tile<kP3W, BF>(patch, base, wr, lane, acc);
            tile_epilogue(wt, acc);
        }
        load_fc0();
        CD_STAMP3();
        CD_STAMP3();
    }
    CD_STAMP3();
    __syncthreads();
    CD_STAMP3();

    // adaptive average pool + flatten: feat[c*16 + bi*4 + bj]
    for (int o = tid; o < kFeat; o += 256) {
        int c = o >> 4, bi = (o >> 2) & 3, bj = o & 3;
        int ys = (bi * Hp) / kPool, ye = ((bi + 1) * Hp + kPool - 1) / kPool;
        int xs = (bj * kW3) / kPool, xe = ((bj + 1) * kW3 + kPool - 1) / kPool;
        float sum = 0.f;
        for (int yy = ys; yy < ye; ++yy)
            for (int xx = xs; xx < xe; ++xx) sum += pooled[(yy * kW3 + xx) * kC + c];
        feat[o] = sum / (float)((ye - ys) * (xe - xs));
    }
    __syncthreads();
    CD_STAMP3();

    // Linear(768, 32): 8 lanes per output, 96 inputs each (interleaved by 16-byte pieces), then a
    // 3-step butterfly
    {
        const f32x4* frow = reinterpret_cast<const f32x4*>(feat) + part;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const f32x4 w = w0[k], f = frow[k * 8];
            sum = fmaf(w[0], f[0], sum); sum = fmaf(w[1], f[1], sum);
            sum = fmaf(w[2], f[2], sum); sum = fmaf(w[3], f[3], sum);
        }
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        if (part == 0) hid[fo] = fmaf(fmaxf(sum + h_bias, 0.f), h_scale, h_shift);
    }
    __syncthreads();
    // Linear(32, 3): one 32-lane group per output, product per lane, 5-step butterfly; then
    // torch.max semantics: first maximum wins; a NaN wins over everything (first NaN)
    if (tid < 64) {
        const int k = tid & 31;                    // lanes 0-31: output 0, lanes 32-63: output 1
        float p0 = l_w01 * hid[k];
        float p2 = l_w2 * hid[k];                  // output 2, computed by both halves
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) { p0 += __shfl_xor(p0, d); p2 += __shfl_xor(p2, d); }
        const float l0 = __shfl(p0, 0) + l_b0, l1 = __shfl(p0, 32) + l_b1, l2 = p2 + l_b2;
        if (tid == 0) {
            const float l[kOut] = {l0, l1, l2};
            float best = l[0];
            int bi = 0;
            for (int o2 = 0; o2 < kOut; ++o2) {
                logits[(int64_t)frame * kOut + o2] = l[o2];
                if (o2 > 0 && !(best != best) && (l[o2] > best || l[o2] != l[o2])) { best = l[o2]; bi = o2; }
            }
            if (maxv) maxv[frame] = best;
            if (argmax) argmax[frame] = bi;
        }
    }
    CD_STAMP3();
}

// The head kernel needs more dynamic LDS than the default limit; the limit is a per-device
// attribute of the kernel, so this runs once per handle with the handle's device current.
int init_conv3_head() {
    CD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_head_kernel<kF32>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    CD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_head_kernel<kBF16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    CD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_head_kernel<kS2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    return CD_OK;
}

int launch_conv3_head(const cd_handle* h, const Workspace& ws, int n, const Geometry& g, float* logits, float* maxv,
                      int64_t* argmax, hipStream_t s) {
    const bool bf = h->dtype == CD_BF16;
    size_t lds = head_lds_bytes(g, bf ? 2 : 4);
    if (lds > (size_t)kLdsBytes) {
        set_error("network input of %d rows needs %zu bytes of LDS in the head kernel (max %d)", g.h, lds, kLdsBytes);
        return CD_EINVAL;
    }
    // K split over the fourth wave (fp32, at most three wave tiles): needs one tile's accumulators
    // (27 x 64 x 16 bytes) of spare LDS, and room for the others in the patch once it is dead
    const int wave_tiles = (g.h3 * kW3 + 15) / 16;
    const size_t part = (size_t)27 * 64 * 16, patch_bytes = (size_t)(g.h2 + 2) * (kW2 + 2) * kC * 4;
    int spare_off = -1;
    if (!bf && wave_tiles <= 3 && wave_tiles * part <= patch_bytes && lds + part <= (size_t)kLdsBytes) {
        spare_off = (int)((lds + 15) / 16 * 4);            // floats, 16-byte aligned
        lds = (size_t)spare_off * 4 + part;
    }
    if (bf)
        hipLaunchKernelGGL(conv3_head_kernel<kBF16>, dim3((unsigned)n), dim3(256), lds, s,
                           reinterpret_cast<const uint16_t*>(ws.act2), (const void*)h->p.w3b, h->p.post + 6 * kC,
                           h->p.fc0w, h->p.fc0p, h->p.fc1w, h->p.fc1b, logits, maxv, argmax, g.h2, g.h3, spare_off);
    else if (h->dtype == CD_BF16_SPLIT)
        hipLaunchKernelGGL(conv3_head_kernel<kS2>, dim3((unsigned)n), dim3(256), lds, s,
                           reinterpret_cast<const uint32_t*>(ws.act2), (const void*)h->p.w3s, h->p.post + 6 * kC,
                           h->p.fc0w, h->p.fc0p, h->p.fc1w, h->p.fc1b, logits, maxv, argmax, g.h2, g.h3, spare_off);
    else
        hipLaunchKernelGGL(conv3_head_kernel<kF32>, dim3((unsigned)n), dim3(256), lds, s, (const float*)ws.act2,
                           (const void*)h->p.w3, h->p.post + 6 * kC, h->p.fc0w, h->p.fc0p, h->p.fc1w, h->p.fc1b,
                           logits, maxv, argmax, g.h2, g.h3, spare_off);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

}  // namespace cd

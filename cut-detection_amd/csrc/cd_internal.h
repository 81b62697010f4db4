// Internal declarations shared by the translation units of libcutdet.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdarg.h>

#include <vector>

#include "../../include/cutdet.h"

namespace cd {

// ---- network geometry (prod_net; input width is fixed at 256) ---------------------------------
constexpr int kC = 48;          // channels of every conv layer's output
constexpr int kInC = 3;         // RGB
constexpr int kNetW = 256;      // network input width
constexpr int kW1 = 85;         // width after block 1  (256 / 3)
constexpr int kW2 = 28;         // width after block 2  (85 / 3)
constexpr int kW3 = 9;          // width after block 3  (28 / 3)
constexpr int kPool = 4;        // adaptive average pool output size
constexpr int kFeat = kC * kPool * kPool;   // 768
constexpr int kHid = 32;
constexpr int kOut = 3;
// block-1 LDS tile: 3 pooled rows of one frame = 255 pool windows = 16 wave tiles
constexpr int kC1Rows = 11;       // input rows held (3 pooled rows need 9 conv rows + halo)
constexpr int kC1RS = 272;        // LDS row stride in floats (columns -1 .. 256 used); 272 = 16 mod 32, see c1_tap
constexpr int kC1PoolRows = 3;
constexpr int kC1Windows = kC1PoolRows * kW1;   // 255
constexpr int kP1W = kW1 + 2;     // padded width of the block-1 output in HBM (87)
constexpr int kP2W = kW2 + 2;     // padded width of the block-2 output in HBM (30)
// K order of block 1.  The 27 taps (ci, ky, kx) plus one zero-weight pad fill 28 k-slots = 7 MFMA
// k-steps x 4 (slot k = 4*step + kq).  In an A-operand LDS read the two kq values of a 32-lane group
// hit the same 16 banks unless their addresses differ by 16 (mod 32) dwords.  With a row stride of
// 16 (mod 32) floats, taps that differ by one row (ky) or one channel plane (11 rows) do; the slots
// are therefore filled with such pairs: 9 pairs (ci,0,kx)/(ci,1,kx), 3 pairs (0,2,kx)/(1,2,kx),
// (2,2,0) with the pad (which reads (2,1,0) with weight 0), and one pair left with a 2-way conflict,
// (2,2,1)/(2,2,2).  Returns ci*100 + ky*10 + kx, negated-minus-one for the pad.
constexpr int c1_tap(int k) {
    const int pair = k >> 1, second = k & 1;
    if (pair < 9) return (pair % 3) * 100 + second * 10 + pair / 3;            // (ci, 0|1, kx)
    if (pair < 12) return second * 100 + 20 + (pair - 9);                      // (0|1, 2, kx)
    if (pair == 12) return second ? -(2 * 100 + 10 + 0) - 1 : 2 * 100 + 20 + 0;  // (2,2,0) + pad at (2,1,0)
    return 2 * 100 + 20 + 1 + second;                                          // (2,2,1), (2,2,2)
}
// Operand-cell frames of the fused block 1 + 2 kernel (split bf16): 8-byte cells {R, G, B, 0} in bfloat16,
// [kCellPad + net_h + kCellPad rows][kCellPitch cells] per frame, pixel (y, x) at row y + kCellPad, cell
// x + kCellPad, everything else zero.  The margin of 4 makes the 44-cell rows a conv-2 tile needs start
// on 16-byte boundaries (cell 36 * tile) and keeps them inside the pitch.
constexpr int kCellPad = 4;
constexpr int kCellPitch = kNetW + kCellPad;     // 260 cells = 2080 bytes per row
inline size_t cell_frame_bytes(int net_h) { return (size_t)(net_h + 2 * kCellPad) * kCellPitch * 8; }
constexpr int kMaxSrcW = 8192;  // widest source frame the resize tables are sized for
constexpr int kMaxNetH = 215;   // tallest network input whose block-2 activation fits the head kernel's LDS
constexpr int kLdsBytes = 160 * 1024;

struct Geometry {               // per network-input height
    int h, h1, h2, h3;          // input rows, rows after each conv block
};
inline Geometry geometry(int h) { Geometry g; g.h = h; g.h1 = h / 3; g.h2 = g.h1 / 3; g.h3 = g.h2 / 3; return g; }
// LDS of the block-3 + head kernel: padded block-2 frame, pooled block-3 tensor, features, hidden units,
// block-3 bias / BatchNorm terms
inline size_t head_lds_bytes(const Geometry& g, int elem_bytes) {
    return (size_t)elem_bytes * (g.h2 + 2) * (kW2 + 2) * kC + sizeof(float) * ((size_t)g.h3 * kW3 * kC + kFeat + kHid + 3 * kC);
}

// ---- resize tables (device) -------------------------------------------------------------------
struct ResizeTables {
    int* sx;  short* ax;   // [w] left sample column; [2w] 11-bit coefficient pairs
    int* sy;  short* by;   // [h] top sample row (unclamped); [2h] coefficient pairs
    int mode;              // 0 bilinear fixed point, 1 pure point sample, 2 exact 2x2 box
    int H0, W0, h, w;      // geometry the tables currently describe (0 = none)
    int affine, xm, xa, ym, ya;   // mode 1: sx[x] = xm*x + xa and sy[y] = ym*y + ya hold for every entry
};

// ---- packed parameters on the device ----------------------------------------------------------
struct DeviceParams {
    float* w1;    // conv1: [9 pieces][64 lanes][4]: B fragments (pieces 0-5), bias / BN scale / BN shift (6-8)
    float* w2;    // conv2 B fragments  [27 steps][3 n-tiles][64 lanes][4]
    float* w3;    // conv3 B fragments  (same layout as w2)
    uint16_t* w1b;  // conv1 B fragments in bf16: [3 ky][3 n-tiles][64 lanes][4] (kx = lane>>4, channel e; zero pads)
    uint16_t* w2b;  // conv2 / conv3 B fragments in bf16: [27 steps][3 n-tiles][64 lanes][4]
    uint16_t* w3b;
    uint32_t* w1x;  // the third term of the same weights, [2][3 n-tiles][64 lanes]: {lo2 row 0, lo2 row 1} and {lo2 row 2, 0}
    uint32_t* w1s;  // conv1 B fragments for split bf16: [3 ky][3 n-tiles][64 lanes][hi(R,G,B),0, lo(R,G,B),0] of w / 255
    uint32_t* w2s;  // conv2 / conv3 B fragments as split bf16: [27 steps][3 n-tiles][64 lanes][hi x 4, lo x 4]
    uint32_t* w3s;
    float* post;  // [3 layers][3][48]: conv bias, BN scale, BN shift
    float* fc0w;  // [32][768]
    float* fc0p;  // [3][32]: bias, BN scale, BN shift
    float* fc1w;  // [3][32]
    float* fc1b;  // [3]
    float* lut;   // [256] k/255 in fp32 (correctly rounded division)
};

}  // namespace cd

namespace cd {
struct Workspace {
    float* act1;            // [frames+1][h1+2][87][48], zero border around each frame
    float* act2;            // [frames+1][h2+2][30][48], zero border around each frame
};
}  // namespace cd

struct cd_handle {
    int device;
    int dtype;
    int opt[CD_OPT_COUNT];     // cd_set_option
    cd::DeviceParams p;
    cd::ResizeTables rt;       // tables of the fused forward (cd_forward_u8 and its variants)
    cd::ResizeTables rt_pre;   // tables of the stand-alone kernels (cd_preprocess, cd_resize_u8): a caller that
                               // alternates a stand-alone resize with a forward on its output keeps both warm
    cd::ResizeTables rt_id;    // tables of the forward on frames that are already net_h x 256 (the second half
                               // of cd_forward_u8_range's resize-then-classify pipeline)
    void* table_mem;        // one allocation behind rt.*
    void* param_mem;        // one allocation behind p.*
    // Activation workspaces.  ws[0] serves the single-batch entry points; cd_forward_u8_range alternates
    // batches between ws[0] on the caller's stream and ws[1] on a stream of its own.
    cd::Workspace ws[2];
    size_t act1_bytes[2], act2_bytes[2];
    int ws_h[2];            // network input height each workspace's borders are laid out for
    // cd_forward_u8_range: side streams (classification of every other batch; stand-alone resize) and
    // the events that fork from / join into the caller's stream and hand resized batches over
    hipStream_t side_cls, side_pre;
    bool own_side_streams;  // false: the device's shared pair (the usual case), not destroyed with the handle
    hipEvent_t ev_fork, ev_join, ev_ready[3], ev_free[3], ev_a1ready[2], ev_a1free[2];
    uint8_t* small;         // resized frames of up to three batches in flight: [3][small_frames][net_h][256][3]
    size_t small_bytes;     // bytes of one of the three slots
    uint8_t* cells;         // split bf16: operand-cell frames of up to three batches in flight (see kCellPad)
    size_t cells_bytes;     // bytes of one of the three slots ((frames + 1) x cell_frame_bytes)
    unsigned* tickets;      // [2048] role tickets of the fused kernel's workgroups, one counter per (CU, SIMD)
    int cells_h;            // network input height the cell frames' zero margins are laid out for
    int64_t* rle_blocks;    // workspace of cd_rle: per-block boundary counts / offsets
    double* rle_sums;       // ... and per-block fp64 sums of the maxima (same number of entries)
    int rle_blocks_cap;
    // optional per-stage timing: HIP event pairs recorded around each stage's launches
    unsigned timing;   // bit i set: launches of stage i are bracketed
    int timing_every;  // ... every timing_every-th launch of the stage (1 = all)
    unsigned timing_seen[CD_STAGE_COUNT];
    std::vector<hipEvent_t> ev_start[CD_STAGE_COUNT], ev_stop[CD_STAGE_COUNT];
    size_t ev_used[CD_STAGE_COUNT];
};

namespace cd {

void set_error(const char* fmt, ...);
#define CD_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            cd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return CD_EHIP;                                                               \
        }                                                                                 \
    } while (0)

// kernel launchers (each queues work on `s` and returns a CD_* code)
int launch_preprocess(const cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                      int64_t frame_stride, int chan_order, float* dst, int nh, int nw, hipStream_t s);
int launch_resize_u8(const cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                     int64_t frame_stride, uint8_t* dst, int nh, int nw, hipStream_t s);
// (the forward kernels read and write the activation buffers of `ws`; `rt` holds the resize tables)
bool resize_cells_supported(const uint8_t* src, int W0, int64_t row_stride, int64_t frame_stride, int nw);
int launch_resize_cells(const cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                        int64_t frame_stride, int chan_order, uint8_t* cells, int nh, int nw, hipStream_t s);
// blocks 1 + 2 fused (split bf16 only): operand-cell frames in, block-2 activations of `ws` out
int launch_conv12_split(const cd_handle* h, const Workspace& ws, const uint8_t* cells, int n, const Geometry& g,
                        hipStream_t s);
int launch_conv1_f32(const cd_handle* h, const Workspace& ws, const float* x, int n, const Geometry& g, hipStream_t s);
int launch_conv1_u8(const cd_handle* h, const Workspace& ws, const ResizeTables& rt, const uint8_t* src, int n, int H0,
                    int W0, int64_t row_stride, int64_t frame_stride, int chan_order, const Geometry& g, hipStream_t s);
int launch_conv2(const cd_handle* h, const Workspace& ws, int n, const Geometry& g, hipStream_t s, bool far_weights = false);
int init_conv3_head();
int init_conv2_split2();     // same, for the two-waves-per-SIMD block-2 kernel of split bf16
int launch_conv3_head(const cd_handle* h, const Workspace& ws, int n, const Geometry& g, float* logits, float* maxv,
                      int64_t* argmax, hipStream_t s);
int launch_rle(const cd_handle* h, const int64_t* cls, const float* maxv, int64_t n, int64_t frame_offset,
               int64_t* end, int64_t* type, int64_t* runlen, int64_t* start, float* mean, double* sum,
               int64_t cap, int64_t* n_seg, hipStream_t s);
int launch_max_argmax(const float* scores, int64_t n, int ncls, float* maxv, int64_t* argmax, hipStream_t s);
int launch_nv12_to_bgr(const uint8_t* y, const uint8_t* uv, int n, int W0, int64_t y_rs, int64_t uv_rs,
                       int64_t y_fs, int64_t uv_fs, const int32_t* y_rows, const int32_t* uv_rows, int n_rows,
                       uint8_t* dst, int chan_order, hipStream_t s);
int launch_eval_accumulate(const float* logits, const int64_t* labels, int64_t n, int ncls, int64_t* correct,
                           int64_t* total, double* loss_sum, hipStream_t s);
int launch_synth(uint8_t* dst, int n, int H0, int W0, const uint8_t* classes, uint32_t seed,
                 int64_t first_index, hipStream_t s);

}  // namespace cd

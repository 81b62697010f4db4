// C ABI of libcutdet.so: handle management, parameter packing, argument checks.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "cd_internal.h"

// Activations of the allocation-free forward (cd_forward_u8_static): one fixed 32 MiB array per
// GPU; each device that runs the library holds its own copy, at its own address, until the process ends.
constexpr size_t kStaticScratchBytes = (size_t)32 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace cd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

namespace {

constexpr char kMagic[8] = {'C', 'D', 'N', 'E', 'T', 'W', '1', '\0'};
constexpr size_t kHeader = 8 + 8 * 4;

struct ConvHost { const float *w, *b, *g, *beta, *mu, *var; };

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// BatchNorm(eval) as the affine y = x*scale + shift, evaluated in fp32 in the order torch's CPU
// kernel uses: invstd = 1/sqrt(var+eps); scale = invstd*gamma; shift = beta - mean*scale.
void bn_affine(const float* g, const float* beta, const float* mu, const float* var, int n, float* scale,
               float* shift) {
    for (int c = 0; c < n; ++c) {
        float invstd = 1.0f / std::sqrt(var[c] + 1e-5f);
        scale[c] = invstd * g[c];
        shift[c] = beta[c] - mu[c] * scale[c];
    }
}

struct ParamOffsets { size_t w1, w2, w3, post, fc0w, fc0p, fc1w, fc1b, lut, w2b, w3b, w1b, w2s, w3s, w1s, w1x; };

// float -> bfloat16 bits, round to nearest even (what v_cvt_pk_bf16_f32 and torch's .bfloat16() do)
static uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Packs everything the kernels read into one host image; `off` receives where each section
// starts (in floats) so the caller can rebase them onto the device allocation.
std::vector<float> pack_params(const float* f, ParamOffsets* off) {
    ConvHost cv[3];
    const float* cur = f;
    int cin = kInC;
    for (int l = 0; l < 3; ++l) {
        cv[l].w = cur; cur += (size_t)kC * cin * 9;
        cv[l].b = cur; cur += kC;
        cv[l].g = cur; cur += kC;
        cv[l].beta = cur; cur += kC;
        cv[l].mu = cur; cur += kC;
        cv[l].var = cur; cur += kC;
        cin = kC;
    }
    const float* fc0w = cur; cur += (size_t)kHid * kFeat;
    const float* fc0b = cur; cur += kHid;
    const float* fc0g = cur; cur += kHid;
    const float* fc0beta = cur; cur += kHid;
    const float* fc0mu = cur; cur += kHid;
    const float* fc0var = cur; cur += kHid;
    const float* fc1w = cur; cur += (size_t)kOut * kHid;
    const float* fc1b = cur;

    std::vector<float> img;
    auto section = [&](size_t n) { size_t at = align_up(img.size(), 64); img.resize(at + n, 0.f); return at; };
    const size_t o_w1 = section(9 * 64 * 4);
    const size_t o_w2 = section(27 * 3 * 64 * 4);
    const size_t o_w3 = section(27 * 3 * 64 * 4);
    const size_t o_post = section(3 * 3 * kC);
    const size_t o_fc0w = section((size_t)kHid * kFeat);
    const size_t o_fc0p = section(3 * kHid);
    const size_t o_fc1w = section(kOut * kHid);
    const size_t o_fc1b = section(kOut);
    const size_t o_lut = section(256);

    // Block 1, everything a lane keeps in registers, as nine 16-byte pieces [piece][lane][4] (a wave
    // reads one contiguous KB per piece).  Pieces 2nt, 2nt+1: B fragments of k-steps 0-3 and 4-6 of
    // channel group nt; lane (kq = lane>>4, col = lane&15) of k-step s holds W[16nt+col][tap c1_tap(4s+kq)].
    // Pieces 6+nt are filled in below with {bias, BN scale, BN shift, 0} of channel 16nt + col.
    for (int s = 0; s < 7; ++s)
        for (int nt = 0; nt < 3; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int tap = c1_tap(4 * s + (lane >> 4)), oc = 16 * nt + (lane & 15);
                img[o_w1 + ((2 * nt + s / 4) * 64 + lane) * 4 + s % 4] =
                    tap >= 0 ? cv[0].w[oc * 27 + (tap / 100) * 9 + ((tap / 10) % 10) * 3 + tap % 10] : 0.f;
            }
    // blocks 2/3: step (ky, kx, g), k-step e covers input channel 16g + 4kq + e
    for (int l = 1; l < 3; ++l) {
        const size_t o = l == 1 ? o_w2 : o_w3;
        for (int kk = 0; kk < 9; ++kk)
            for (int g = 0; g < 3; ++g)
                for (int nt = 0; nt < 3; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            int ci = 16 * g + 4 * (lane >> 4) + e, oc = 16 * nt + (lane & 15);
                            img[o + ((((size_t)(kk * 3 + g) * 3 + nt) * 64 + lane) * 4) + e] =
                                cv[l].w[((size_t)oc * kC + ci) * 9 + kk];
                        }
    }
    // the same fragments in bf16 (round to nearest even), two per float slot of the image
    const size_t o_w2b = section(27 * 3 * 64 * 2), o_w3b = section(27 * 3 * 64 * 2);
    for (int l = 1; l < 3; ++l) {
        const size_t src = l == 1 ? o_w2 : o_w3, dst = l == 1 ? o_w2b : o_w3b;
        uint16_t* d = reinterpret_cast<uint16_t*>(&img[dst]);
        for (size_t k = 0; k < (size_t)27 * 3 * 64 * 4; ++k) d[k] = bf16_rne(img[src + k]);
    }
    // the same fragments as split bf16, hi = bf16(w), lo = bf16(w - hi): 16 bytes per lane, {hi x 4, lo x 4}.
    // The kernels build the two B operands {hi x 4, hi x 4} and {lo x 4, lo x 4} (each half meets one half of
    // the activation cell) in registers: stored that way the weights would be twice the bytes, and block 2's
    // 83 KB already stream from L2 once per wave tile.
#ifdef CD_S2_WDUP
    const size_t o_w2s = section(27 * 3 * 64 * 8), o_w3s = section(27 * 3 * 64 * 8);
#else
    const size_t o_w2s = section(27 * 3 * 64 * 4), o_w3s = section(27 * 3 * 64 * 4);
#endif
    for (int l = 1; l < 3; ++l) {
        const size_t src = l == 1 ? o_w2 : o_w3, dst = l == 1 ? o_w2s : o_w3s;
        uint16_t* d = reinterpret_cast<uint16_t*>(&img[dst]);
        for (size_t f = 0; f < (size_t)27 * 3; ++f)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const float w = img[src + (f * 64 + lane) * 4 + e];
                    const uint16_t hi = bf16_rne(w);
                    const uint32_t hb = (uint32_t)hi << 16;
                    float hf;
                    std::memcpy(&hf, &hb, 4);
                    const uint16_t lo = bf16_rne(w - hf);
#ifdef CD_S2_WDUP
                    uint16_t* ph = d + (f * 2 * 64 + lane) * 8, *pl = d + ((f * 2 + 1) * 64 + lane) * 8;
                    ph[e] = ph[4 + e] = hi;
                    pl[e] = pl[4 + e] = lo;
#else
                    uint16_t* cell = d + (f * 64 + lane) * 8;
                    cell[e] = hi;
                    cell[4 + e] = lo;
#endif
                }
    }
    // block 1 in bf16: fragment (ky, nt), lane (kx = lane>>4, col): W[16nt+col][ci = e][ky][kx], zero for kx = 3 / e = 3
    const size_t o_w1b = section(9 * 64 * 2);
    {
        uint16_t* d = reinterpret_cast<uint16_t*>(&img[o_w1b]);
        for (int ky = 0; ky < 3; ++ky)
            for (int nt = 0; nt < 3; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int kx = lane >> 4, oc = 16 * nt + (lane & 15);
                        const float v = (kx < 3 && e < 3) ? cv[0].w[oc * 27 + e * 9 + ky * 3 + kx] : 0.f;
                        d[((ky * 3 + nt) * 64 + lane) * 4 + e] = bf16_rne(v);
                    }
    }
    // block 1 for split bf16 on u8 input: the bytes enter the matrix core as they are, so the weights carry
    // the 1/255: w' = w / 255 (fp32), fragment (ky, nt), lane (kx = lane>>4, col): {hi(w'[ci=0..2]), 0, lo(...), 0}
    const size_t o_w1s = section(9 * 64 * 4);
    {
        uint16_t* d = reinterpret_cast<uint16_t*>(&img[o_w1s]);
        for (int ky = 0; ky < 3; ++ky)
            for (int nt = 0; nt < 3; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int kx = lane >> 4, oc = 16 * nt + (lane & 15);
                        const float w = (kx < 3 && e < 3) ? cv[0].w[oc * 27 + e * 9 + ky * 3 + kx] / 255.0f : 0.f;
                        const uint16_t hi = bf16_rne(w);
                        const uint32_t hb = (uint32_t)hi << 16;
                        float hf;
                        std::memcpy(&hf, &hb, 4);
                        uint16_t* cell = d + ((ky * 3 + nt) * 64 + lane) * 8;
                        cell[e] = hi;
                        cell[4 + e] = bf16_rne(w - hf);
                    }
    }
    // the third term of the same weights: hi + lo + lo2 == w' exactly (3 x 8 mantissa bits; the remainder after each
    // term is computed exactly in fp32).  Fragment (0, nt): {lo2 of kernel row 0, lo2 of kernel row 1}; fragment
    // (1, nt): {lo2 of kernel row 2, 0} - the operand they meet holds the pixels of two kernel rows
    const size_t o_w1x = section(6 * 64 * 4);
    {
        uint16_t* d = reinterpret_cast<uint16_t*>(&img[o_w1x]);
        const uint16_t* two = reinterpret_cast<const uint16_t*>(&img[o_w1s]);
        auto as_float = [](uint16_t b) { const uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f; };
        std::memset(d, 0, (size_t)6 * 64 * 16);
        for (int ky = 0; ky < 3; ++ky)
            for (int nt = 0; nt < 3; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int kx = lane >> 4, oc = 16 * nt + (lane & 15);
                        const float w = (kx < 3 && e < 3) ? cv[0].w[oc * 27 + e * 9 + ky * 3 + kx] / 255.0f : 0.f;
                        const uint16_t* c2 = two + ((ky * 3 + nt) * 64 + lane) * 8;
                        const float rest = (w - as_float(c2[e])) - as_float(c2[4 + e]);
                        d[(((ky >> 1) * 3 + nt) * 64 + lane) * 8 + 4 * (ky & 1) + e] = bf16_rne(rest);
                    }
    }
    for (int l = 0; l < 3; ++l) {
        float* p = &img[o_post + (size_t)l * 3 * kC];
        std::memcpy(p, cv[l].b, kC * sizeof(float));
        bn_affine(cv[l].g, cv[l].beta, cv[l].mu, cv[l].var, kC, p + kC, p + 2 * kC);
    }
    for (int nt = 0; nt < 3; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
            const int c = 16 * nt + (lane & 15);
            float* d = &img[o_w1 + ((6 + nt) * 64 + lane) * 4];
            d[0] = img[o_post + c]; d[1] = img[o_post + kC + c]; d[2] = img[o_post + 2 * kC + c];
        }
    std::memcpy(&img[o_fc0w], fc0w, (size_t)kHid * kFeat * sizeof(float));
    std::memcpy(&img[o_fc0p], fc0b, kHid * sizeof(float));
    bn_affine(fc0g, fc0beta, fc0mu, fc0var, kHid, &img[o_fc0p + kHid], &img[o_fc0p + 2 * kHid]);
    std::memcpy(&img[o_fc1w], fc1w, kOut * kHid * sizeof(float));
    std::memcpy(&img[o_fc1b], fc1b, kOut * sizeof(float));
    for (int k = 0; k < 256; ++k) img[o_lut + k] = (float)k / 255.0f;   // data.py:227 divides, it does not scale

    *off = ParamOffsets{o_w1, o_w2, o_w3, o_post, o_fc0w, o_fc0p, o_fc1w, o_fc1b, o_lut, o_w2b, o_w3b, o_w1b, o_w2s, o_w3s, o_w1s, o_w1x};
    return img;
}

// One axis of OpenCV's 8U INTER_LINEAR tables (float/double mix as in resize.cpp).
void axis_tables(int src_n, int dst_n, bool zero_at_border, std::vector<int>& s, std::vector<short>& c) {
    const double inv = (double)dst_n / src_n, scale = 1.0 / inv;
    s.resize(dst_n);
    c.resize(2 * (size_t)dst_n);
    for (int d = 0; d < dst_n; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int i = (int)std::floor(f);
        f -= (float)i;
        if (zero_at_border) {
            if (i < 0) { f = 0.f; i = 0; }
            if (i >= src_n - 1) { f = 0.f; i = src_n - 1; }
        }
        long c0 = lrintf((1.f - f) * 2048.f), c1 = lrintf(f * 2048.f);
        s[d] = i;
        c[2 * d] = (short)std::min(32767l, std::max(-32768l, c0));
        c[2 * d + 1] = (short)std::min(32767l, std::max(-32768l, c1));
    }
}

int ensure_tables(ResizeTables& rt, int H0, int W0, int nh, int nw, hipStream_t s) {
    if (rt.H0 == H0 && rt.W0 == W0 && rt.h == nh && rt.w == nw) return CD_OK;
    if (H0 < 1 || W0 < 1 || nh < 1 || nw < 1 || W0 > kMaxSrcW || H0 > kMaxSrcW || nh > kMaxSrcW || nw > kMaxSrcW) {
        set_error("resize %dx%d -> %dx%d outside the supported range (1..%d per side)", W0, H0, nw, nh, kMaxSrcW);
        return CD_EINVAL;
    }
    std::vector<int> sx, sy;
    std::vector<short> ax, by;
    axis_tables(W0, nw, true, sx, ax);
    axis_tables(H0, nh, false, sy, by);
    int mode = 0;
    if (H0 == 2 * nh && W0 == 2 * nw) mode = 2;
    else {
        bool point = true;
        for (int x = 0; x < nw; ++x) point &= ax[2 * x + 1] == 0 && ax[2 * x] == 2048;
        for (int y = 0; y < nh; ++y) point &= by[2 * y + 1] == 0 && by[2 * y] == 2048 && sy[y] >= 0 && sy[y] < H0;
        if (point) mode = 1;
    }
    // Tables of the previous geometry may still be in use by queued kernels.
    CD_HIP(hipStreamSynchronize(s));
    rt.H0 = rt.W0 = rt.h = rt.w = 0;      // nothing valid until every table below is in place
    CD_HIP(hipMemcpy(rt.sx, sx.data(), sx.size() * sizeof(int), hipMemcpyHostToDevice));
    CD_HIP(hipMemcpy(rt.ax, ax.data(), ax.size() * sizeof(short), hipMemcpyHostToDevice));
    CD_HIP(hipMemcpy(rt.sy, sy.data(), sy.size() * sizeof(int), hipMemcpyHostToDevice));
    CD_HIP(hipMemcpy(rt.by, by.data(), by.size() * sizeof(short), hipMemcpyHostToDevice));
    rt.affine = 0;
    if (mode == 1 && nw >= 2 && nh >= 2) {
        const int xm = sx[1] - sx[0], xa = sx[0], ym = sy[1] - sy[0], ya = sy[0];
        bool ok = true;
        for (int x = 0; x < nw; ++x) ok &= sx[x] == xm * x + xa;
        for (int y = 0; y < nh; ++y) ok &= sy[y] == ym * y + ya;
        if (ok) { rt.affine = 1; rt.xm = xm; rt.xa = xa; rt.ym = ym; rt.ya = ya; }
    }
    rt.mode = mode; rt.H0 = H0; rt.W0 = W0; rt.h = nh; rt.w = nw;
    return CD_OK;
}

// Activations are stored with a one-pixel zero border per frame, plus one frame of slack at the
// end so that clamped reads of partial tiles stay inside the allocation.  The whole buffer is
// zeroed once here; the kernels only ever write interior cells, so the border stays zero.
int ensure_workspace(cd_handle* h, int k, int n, const Geometry& g, hipStream_t s) {
    const size_t need1 = (size_t)(n + 1) * (g.h1 + 2) * (kW1 + 2) * kC * sizeof(float);
    const size_t need2 = (size_t)(n + 1) * (g.h2 + 2) * (kW2 + 2) * kC * sizeof(float);
    Workspace& w = h->ws[k];
    const bool same_shape = h->ws_h[k] == g.h;
    if (same_shape && need1 <= h->act1_bytes[k] && need2 <= h->act2_bytes[k]) return CD_OK;
    CD_HIP(hipStreamSynchronize(s));
    if (k == 1 && h->side_cls) CD_HIP(hipStreamSynchronize(h->side_cls));   // the stream that last used ws[1]
    if (need1 > h->act1_bytes[k]) {
        if (w.act1) CD_HIP(hipFree(w.act1));
        w.act1 = nullptr; h->act1_bytes[k] = 0;
        CD_HIP(hipMalloc(reinterpret_cast<void**>(&w.act1), need1));
        h->act1_bytes[k] = need1;
    }
    if (need2 > h->act2_bytes[k]) {
        if (w.act2) CD_HIP(hipFree(w.act2));
        w.act2 = nullptr; h->act2_bytes[k] = 0;
        CD_HIP(hipMalloc(reinterpret_cast<void**>(&w.act2), need2));
        h->act2_bytes[k] = need2;
    }
    // a different input height moves the borders: re-zero everything
    CD_HIP(hipMemsetAsync(w.act1, 0, h->act1_bytes[k], s));
    CD_HIP(hipMemsetAsync(w.act2, 0, h->act2_bytes[k], s));
    // The zeroing is complete before this returns: the next call may come on another stream, which
    // nothing would order behind a memset still queued on this one.
    CD_HIP(hipStreamSynchronize(s));
    h->ws_h[k] = g.h;
    return CD_OK;
}

// ---- activations of the allocation-free forward ----------------------------------------------
// `scratch` (file scope, above): host code never uses the symbol's name as an address.  cd_create()
// asks the runtime for the array's device address, once per device with that device current, and
// everything below works with that pointer.
constexpr int kMaxDevices = 64;
struct StaticScratch {
    unsigned char* ptr = nullptr;   // device address of `scratch` on this GPU
    int ws_h = 0;                   // network input height its zero borders are laid out for
    int ws_es = 0;                  // ... and the activation element size (2: bf16 cells, 4: fp32 / split bf16)
};
StaticScratch g_static[kMaxDevices];
std::mutex g_static_mutex;

// Layout of the array for one network input height: block-1 activations at offset 0, block-2
// activations at a fixed offset, both sized for the most frames that fit (plus the slack frame), so
// that the zero borders sit at the same addresses whatever the batch size.
struct StaticLayout { int max_frames; size_t act2_off; };
StaticLayout static_layout(const Geometry& g) {
    const size_t f1 = (size_t)(g.h1 + 2) * (kW1 + 2) * kC * sizeof(float);
    const size_t f2 = (size_t)(g.h2 + 2) * (kW2 + 2) * kC * sizeof(float);
    long m = (long)(kStaticScratchBytes / (f1 + f2));
    while (m > 0 && align_up((size_t)m * f1, 256) + (size_t)m * f2 > kStaticScratchBytes) --m;
    return StaticLayout{(int)(m - 1), m > 0 ? align_up((size_t)m * f1, 256) : 0};
}

int init_static_scratch(int device) {
    if (device >= kMaxDevices) return CD_OK;       // such a device has no allocation-free entry point
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (g_static[device].ptr) return CD_OK;
    void* p = nullptr;
    CD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_static[device].ptr = static_cast<unsigned char*>(p);
    return CD_OK;
}

int check_net_shape(const cd_handle* h, int n, int net_h, int net_w, Geometry* g) {
    if (n < 1) { set_error("batch of %d frames", n); return CD_EINVAL; }
    if (net_w != kNetW) { set_error("network input width must be %d, got %d", kNetW, net_w); return CD_EINVAL; }
    *g = geometry(net_h);
    if (g->h3 < 1 || net_h > kMaxNetH || head_lds_bytes(*g, h->dtype == CD_BF16 ? 2 : 4) > (size_t)kLdsBytes) {
        set_error("network input height %d unsupported (27..%d rows)", net_h, kMaxNetH);
        return CD_EINVAL;
    }
    return CD_OK;
}

// Brackets the launches issued during its lifetime with an event pair when timing is on.
struct StageTimer {
    cd_handle* h;
    int stage;
    hipStream_t s;
    bool live = false;
    StageTimer(cd_handle* h_, int stage_, hipStream_t s_) : h(h_), stage(stage_), s(s_) {
        if (!((h->timing >> stage) & 1u)) return;
        if (h->timing_seen[stage]++ % (unsigned)(h->timing_every > 0 ? h->timing_every : 1)) return;
        size_t k = h->ev_used[stage];
        if (k == h->ev_start[stage].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess) return;
            if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
            h->ev_start[stage].push_back(a);
            h->ev_stop[stage].push_back(b);
        }
        live = hipEventRecord(h->ev_start[stage][k], s) == hipSuccess;
    }
    ~StageTimer() {
        if (!live) return;
        size_t k = h->ev_used[stage];
        if (hipEventRecord(h->ev_stop[stage][k], s) == hipSuccess) h->ev_used[stage] = k + 1;
    }
};

// The three kernels of one batch on stream s, activations in `ws`, resize tables `rt` (already on the
// device for this geometry).  `timed`: bracket the stages with events when the handle's timing is on.
int forward_u8_on(cd_handle* h, const Workspace& ws, const ResizeTables& rt, const uint8_t* src, int n, int H0, int W0,
                  int64_t row_stride, int64_t frame_stride, int chan_order, const Geometry& g, float* logits, float* maxv,
                  int64_t* argmax, hipStream_t s, bool timed) {
    int rc;
    if (!timed) {
        if ((rc = launch_conv1_u8(h, ws, rt, src, n, H0, W0, row_stride, frame_stride, chan_order, g, s))) return rc;
        if ((rc = launch_conv2(h, ws, n, g, s))) return rc;
        return launch_conv3_head(h, ws, n, g, logits, maxv, argmax, s);
    }
    {
        StageTimer t(h, CD_STAGE_CONV1, s);
        if ((rc = launch_conv1_u8(h, ws, rt, src, n, H0, W0, row_stride, frame_stride, chan_order, g, s))) return rc;
    }
    { StageTimer t(h, CD_STAGE_CONV2, s); if ((rc = launch_conv2(h, ws, n, g, s))) return rc; }
    StageTimer t(h, CD_STAGE_CONV3_HEAD, s);
    return launch_conv3_head(h, ws, n, g, logits, maxv, argmax, s);
}

// Whether the resize H0 x W0 -> nh x nw is a pure point sample (every tap weight 0 or 1 and in range):
// then the fused first block gathers single source pixels and a separate resize pass gains nothing.
bool resize_is_point_sample(int H0, int W0, int nh, int nw) {
    std::vector<int> sx, sy;
    std::vector<short> ax, by;
    axis_tables(W0, nw, true, sx, ax);
    axis_tables(H0, nh, false, sy, by);
    if (H0 == 2 * nh && W0 == 2 * nw) return false;
    bool point = true;
    for (int x = 0; x < nw; ++x) point &= ax[2 * x + 1] == 0 && ax[2 * x] == 2048;
    for (int y = 0; y < nh; ++y) point &= by[2 * y + 1] == 0 && by[2 * y] == 2048 && sy[y] >= 0 && sy[y] < H0;
    return point;
}

// Defaults of the per-handle options of cd_forward_u8_range (cd_set_option), from the environment:
// CUTDET_RANGE_MODE (0 / 1), CUTDET_FUSED12 (0 / 1), CUTDET_RANGE_ONE_STREAM (0 / 1).
// Timing experiments only (builds with -DCD_EXP_SKIP): CUTDET_SKIP is a mask of the range call's stages to
// leave out (1 resize pass, 2 block 1, 4 block 2, 8 block 3 + head); the results are then garbage.
#ifdef CD_EXP_SKIP
static int skip_mask() { static const int m = [] { const char* e = std::getenv("CUTDET_SKIP"); return e ? std::atoi(e) : 0; }(); return m; }
#define CD_SKIP_IF(bit) if (!(skip_mask() & (bit)))
#else
#define CD_SKIP_IF(bit)
#endif

int env_option(const char* name, int fallback) {
    const char* e = std::getenv(name);
    return e && (e[0] == '0' || e[0] == '1') && !e[1] ? e[0] - '0' : fallback;
}

// The two side streams of cd_forward_u8_range exist once per device and are shared by every handle on it: a
// process has few hardware queues (4 by default) and streams are dealt onto them in creation order, so a
// pair per handle soon puts two streams of ONE call on the same queue (measured: the bench's second
// handle 3 % slower than the same work in a process of its own).  They live until the process ends.
struct SideStreams { hipStream_t cls = nullptr, pre = nullptr; };
SideStreams g_side[kMaxDevices];
std::mutex g_side_mutex;

static int create_side_streams(hipStream_t* a, hipStream_t* b) {
#ifdef CD_EXP_ENV        // measurement build: stream priorities from the environment (-1 high, 0 normal, 1 low)
    const char* pa = std::getenv("CUTDET_X_CLS_PRIO");
    const char* pb = std::getenv("CUTDET_X_PRE_PRIO");
    CD_HIP(hipStreamCreateWithPriority(a, hipStreamNonBlocking, pa ? std::atoi(pa) : 0));
    if (hipStreamCreateWithPriority(b, hipStreamNonBlocking, pb ? std::atoi(pb) : 0) != hipSuccess) {
#else
    CD_HIP(hipStreamCreateWithFlags(a, hipStreamNonBlocking));
    if (hipStreamCreateWithFlags(b, hipStreamNonBlocking) != hipSuccess) {
#endif
        (void)hipStreamDestroy(*a);
        set_error("cd_forward_u8_range: cannot create a stream");
        return CD_EHIP;
    }
    return CD_OK;
}

// Streams and events of cd_forward_u8_range, created on first use (with the handle's device current).
int ensure_range_state(cd_handle* h) {
    if (h->side_cls) return CD_OK;
    hipStream_t a = nullptr, b = nullptr;
    if (h->device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(g_side_mutex);
        SideStreams& ss = g_side[h->device];
        if (!ss.cls) {
            int rc = create_side_streams(&a, &b);
            if (rc) return rc;
            ss.cls = a;
            ss.pre = b;
        }
        a = ss.cls;
        b = ss.pre;
        h->own_side_streams = false;
    } else {
        int rc = create_side_streams(&a, &b);
        if (rc) return rc;
        h->own_side_streams = true;
    }
    hipEvent_t* ev[12] = {&h->ev_fork, &h->ev_join, &h->ev_ready[0], &h->ev_ready[1], &h->ev_ready[2],
                          &h->ev_free[0], &h->ev_free[1], &h->ev_free[2], &h->ev_a1ready[0], &h->ev_a1ready[1],
                          &h->ev_a1free[0], &h->ev_a1free[1]};
    for (hipEvent_t* e : ev)
        if (!*e) CD_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    h->side_pre = b;
    h->side_cls = a;
    return CD_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace
}  // namespace cd

using namespace cd;

extern "C" {

int cd_version(void) { return 3; }

const char* cd_last_error(void) { return cd::g_err; }

int cd_create(cd_handle** out, const void* weights_blob, size_t nbytes, int device, int dtype) {
    if (!out || !weights_blob) { set_error("cd_create: null argument"); return CD_EINVAL; }
    *out = nullptr;
    if (dtype != CD_F32 && dtype != CD_BF16 && dtype != CD_BF16_SPLIT) {
        set_error("cd_create: dtype %d not supported (0 = f32, 1 = bf16, 2 = split bf16)", dtype);
        return CD_EINVAL;
    }
    const unsigned char* b = static_cast<const unsigned char*>(weights_blob);
    if (nbytes < kHeader || std::memcmp(b, kMagic, 8) != 0) { set_error("cd_create: not a weight blob"); return CD_EINVAL; }
    uint32_t hd[8];
    std::memcpy(hd, b + 8, sizeof hd);
    const uint32_t want_floats = (kC * kInC * 9 + 5 * kC) + 2 * (kC * kC * 9 + 5 * kC) +
                                 (kHid * kFeat + 5 * kHid) + (kOut * kHid + kOut);
    if (hd[0] != 1 || hd[1] != 3 || hd[2] != (uint32_t)kC || hd[3] != (uint32_t)kInC || hd[4] != (uint32_t)kPool ||
        hd[5] != (uint32_t)kHid || hd[6] != (uint32_t)kOut || hd[7] != want_floats ||
        nbytes != kHeader + (size_t)want_floats * 4) {
        set_error("cd_create: blob is not a version-1 prod_net (3 conv x 48 ch, pool 4, 32 hidden, 3 classes)");
        return CD_EINVAL;
    }
    int ndev = 0;
    CD_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { set_error("cd_create: device %d of %d", device, ndev); return CD_EINVAL; }
    DeviceGuard guard(device);
    if (!guard.ok) { set_error("cd_create: cannot select device %d", device); return CD_EHIP; }

    std::vector<float> f(want_floats);
    std::memcpy(f.data(), b + kHeader, (size_t)want_floats * 4);
    cd_handle* h = new (std::nothrow) cd_handle();
    if (!h) { set_error("cd_create: out of host memory"); return CD_ENOMEM; }
    h->device = device;
    h->dtype = dtype;
    h->opt[CD_OPT_RANGE_BY_STAGE] = env_option("CUTDET_RANGE_MODE", 0);
    h->opt[CD_OPT_FUSE_BLOCKS12] = env_option("CUTDET_FUSED12", 0);
    h->opt[CD_OPT_RANGE_ONE_STREAM] = env_option("CUTDET_RANGE_ONE_STREAM", 0);
    h->opt[CD_OPT_SPLIT_TWO_WAVES] = env_option("CUTDET_S2_TWO_WAVES", 0);
    h->opt[CD_OPT_RESIZE_ROWS] = env_option("CUTDET_RESIZE_ROWS", 0);
    h->opt[CD_OPT_F32_BLOCK1_FP32_MFMA] = env_option("CUTDET_F32_BLOCK1_FP32_MFMA", 0);
    ParamOffsets off;
    std::vector<float> img = pack_params(f.data(), &off);
    auto fail = [&](int code) { cd_destroy(h); return code; };
#define CD_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return fail(CD_EHIP); } } while (0)
    CD_TRY(hipMalloc(&h->param_mem, img.size() * sizeof(float)));
    CD_TRY(hipMemcpy(h->param_mem, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
    float* base = static_cast<float*>(h->param_mem);
    h->p.w1 = base + off.w1; h->p.w2 = base + off.w2; h->p.w3 = base + off.w3; h->p.post = base + off.post;
    h->p.fc0w = base + off.fc0w; h->p.fc0p = base + off.fc0p; h->p.fc1w = base + off.fc1w;
    h->p.fc1b = base + off.fc1b; h->p.lut = base + off.lut;
    h->p.w1b = reinterpret_cast<uint16_t*>(base + off.w1b);
    h->p.w2b = reinterpret_cast<uint16_t*>(base + off.w2b); h->p.w3b = reinterpret_cast<uint16_t*>(base + off.w3b);
    h->p.w1s = reinterpret_cast<uint32_t*>(base + off.w1s);
    h->p.w1x = reinterpret_cast<uint32_t*>(base + off.w1x);
    h->p.w2s = reinterpret_cast<uint32_t*>(base + off.w2s); h->p.w3s = reinterpret_cast<uint32_t*>(base + off.w3s);

    const size_t t_int = align_up((size_t)kMaxSrcW * sizeof(int), 256);
    const size_t t_short = align_up((size_t)2 * kMaxSrcW * sizeof(short), 256);
    CD_TRY(hipMalloc(&h->table_mem, 3 * (2 * t_int + 2 * t_short)));
    char* t = static_cast<char*>(h->table_mem);
    for (ResizeTables* rt : {&h->rt, &h->rt_pre, &h->rt_id}) {
        rt->sx = reinterpret_cast<int*>(t); t += t_int;
        rt->sy = reinterpret_cast<int*>(t); t += t_int;
        rt->ax = reinterpret_cast<short*>(t); t += t_short;
        rt->by = reinterpret_cast<short*>(t); t += t_short;
    }
    h->rle_blocks_cap = 1 << 16;   // x 4096 frames per block
    CD_TRY(hipMalloc(reinterpret_cast<void**>(&h->rle_blocks), (size_t)h->rle_blocks_cap * sizeof(int64_t)));
    CD_TRY(hipMalloc(reinterpret_cast<void**>(&h->rle_sums), (size_t)h->rle_blocks_cap * sizeof(double)));
    CD_TRY(hipMalloc(reinterpret_cast<void**>(&h->tickets), 2048 * sizeof(unsigned)));
    CD_TRY(hipMemset(h->tickets, 0, 2048 * sizeof(unsigned)));
#undef CD_TRY
    if (init_conv3_head() != CD_OK || init_conv2_split2() != CD_OK) return fail(CD_EHIP);
    if (init_static_scratch(device) != CD_OK) return fail(CD_EHIP);
    *out = h;
    return CD_OK;
}

void cd_destroy(cd_handle* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    if (h->param_mem) (void)hipFree(h->param_mem);
    if (h->table_mem) (void)hipFree(h->table_mem);
    for (int k = 0; k < 2; ++k) {
        if (h->ws[k].act1) (void)hipFree(h->ws[k].act1);
        if (h->ws[k].act2) (void)hipFree(h->ws[k].act2);
    }
    if (h->small) (void)hipFree(h->small);
    if (h->cells) (void)hipFree(h->cells);
    if (h->own_side_streams) {
        if (h->side_cls) (void)hipStreamDestroy(h->side_cls);
        if (h->side_pre) (void)hipStreamDestroy(h->side_pre);
    }
    for (hipEvent_t e : {h->ev_fork, h->ev_join, h->ev_ready[0], h->ev_ready[1], h->ev_ready[2], h->ev_free[0],
                         h->ev_free[1], h->ev_free[2], h->ev_a1ready[0], h->ev_a1ready[1], h->ev_a1free[0],
                         h->ev_a1free[1]})
        if (e) (void)hipEventDestroy(e);
    if (h->rle_blocks) (void)hipFree(h->rle_blocks);
    if (h->rle_sums) (void)hipFree(h->rle_sums);
    if (h->tickets) (void)hipFree(h->tickets);
    for (int st = 0; st < CD_STAGE_COUNT; ++st) {
        for (hipEvent_t e : h->ev_start[st]) (void)hipEventDestroy(e);
        for (hipEvent_t e : h->ev_stop[st]) (void)hipEventDestroy(e);
    }
    delete h;
}

int cd_reserve(cd_handle* h, int max_frames, int net_h) {
    if (!h) { set_error("cd_reserve: null handle"); return CD_EINVAL; }
    Geometry g;
    int rc = check_net_shape(h, max_frames, net_h, kNetW, &g);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    return ensure_workspace(h, 0, max_frames, g, nullptr);
}

static int check_src(const char* who, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                     int64_t frame_stride, int chan_order) {
    if (!src || n < 1) { set_error("%s: null source or empty batch", who); return CD_EINVAL; }
    if (chan_order != CD_BGR && chan_order != CD_RGB) { set_error("%s: chan_order %d", who, chan_order); return CD_EINVAL; }
    if (H0 < 1 || W0 < 1 || row_stride < (int64_t)W0 * 3 || frame_stride < row_stride * H0) {
        set_error("%s: strides (%lld, %lld) too small for %dx%d frames", who, (long long)row_stride,
                  (long long)frame_stride, W0, H0);
        return CD_EINVAL;
    }
    return CD_OK;
}

int cd_preprocess(cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                  int64_t frame_stride, int chan_order, float* dst, int net_h, int net_w, void* stream) {
    if (!h || !dst) { set_error("cd_preprocess: null argument"); return CD_EINVAL; }
    int rc = check_src("cd_preprocess", src, n, H0, W0, row_stride, frame_stride, chan_order);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = ensure_tables(h->rt_pre, H0, W0, net_h, net_w, s))) return rc;
    StageTimer t(h, CD_STAGE_PREPROCESS, s);
    return launch_preprocess(h, src, n, H0, W0, row_stride, frame_stride, chan_order, dst, net_h, net_w, s);
}

int cd_nv12_to_bgr(cd_handle* h, const uint8_t* y, const uint8_t* uv, int n, int W0, int64_t y_row_stride,
                   int64_t uv_row_stride, int64_t y_frame_stride, int64_t uv_frame_stride, const int32_t* y_rows,
                   const int32_t* uv_rows, int n_rows, uint8_t* dst, int chan_order, void* stream) {
    if (!h || !y || !uv || !dst || n < 1 || n > 65535 || n_rows < 1 || n_rows > 65535 || W0 < 2 || (W0 & 1)) {
        set_error("cd_nv12_to_bgr: null argument, empty batch, more than 65535 frames or rows, or an odd width");
        return CD_EINVAL;
    }
    if (chan_order != CD_BGR && chan_order != CD_RGB) { set_error("cd_nv12_to_bgr: chan_order %d", chan_order); return CD_EINVAL; }
    if (y_row_stride < W0 || uv_row_stride < W0) {
        set_error("cd_nv12_to_bgr: row strides (%lld, %lld) smaller than the %d-byte rows", (long long)y_row_stride,
                  (long long)uv_row_stride, W0);
        return CD_EINVAL;
    }
    DeviceGuard guard(h->device);
    return launch_nv12_to_bgr(y, uv, n, W0, y_row_stride, uv_row_stride, y_frame_stride, uv_frame_stride, y_rows,
                              uv_rows, n_rows, dst, chan_order, static_cast<hipStream_t>(stream));
}

int cd_resize_u8(cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                 int64_t frame_stride, uint8_t* dst, int net_h, int net_w, void* stream) {
    if (!h || !dst) { set_error("cd_resize_u8: null argument"); return CD_EINVAL; }
    int rc = check_src("cd_resize_u8", src, n, H0, W0, row_stride, frame_stride, CD_BGR);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = ensure_tables(h->rt_pre, H0, W0, net_h, net_w, s))) return rc;
    StageTimer t(h, CD_STAGE_PREPROCESS, s);
    return launch_resize_u8(h, src, n, H0, W0, row_stride, frame_stride, dst, net_h, net_w, s);
}

int cd_forward(cd_handle* h, const float* x, int n, int net_h, int net_w, float* logits, float* maxv,
               int64_t* argmax, void* stream) {
    if (!h || !x || !logits) { set_error("cd_forward: null argument"); return CD_EINVAL; }
    Geometry g;
    int rc = check_net_shape(h, n, net_h, net_w, &g);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = ensure_workspace(h, 0, n, g, s))) return rc;
    { StageTimer t(h, CD_STAGE_CONV1, s); if ((rc = launch_conv1_f32(h, h->ws[0], x, n, g, s))) return rc; }
    { StageTimer t(h, CD_STAGE_CONV2, s); if ((rc = launch_conv2(h, h->ws[0], n, g, s))) return rc; }
    StageTimer t(h, CD_STAGE_CONV3_HEAD, s);
    return launch_conv3_head(h, h->ws[0], n, g, logits, maxv, argmax, s);
}

int cd_forward_u8(cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                  int64_t frame_stride, int chan_order, int net_h, int net_w, float* logits, float* maxv,
                  int64_t* argmax, void* stream) {
    if (!h || !logits) { set_error("cd_forward_u8: null argument"); return CD_EINVAL; }
    int rc = check_src("cd_forward_u8", src, n, H0, W0, row_stride, frame_stride, chan_order);
    if (rc) return rc;
    Geometry g;
    if ((rc = check_net_shape(h, n, net_h, net_w, &g))) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = ensure_tables(h->rt, H0, W0, net_h, net_w, s))) return rc;
    if ((rc = ensure_workspace(h, 0, n, g, s))) return rc;
    return forward_u8_on(h, h->ws[0], h->rt, src, n, H0, W0, row_stride, frame_stride, chan_order, g, logits, maxv,
                         argmax, s, true);
}

int cd_forward_u8_static(cd_handle* h, const uint8_t* src, int n, int H0, int W0, int64_t row_stride,
                         int64_t frame_stride, int chan_order, int net_h, int net_w, float* logits, float* maxv,
                         int64_t* argmax, void* stream) {
    if (!h || !logits) { set_error("cd_forward_u8_static: null argument"); return CD_EINVAL; }
    int rc = check_src("cd_forward_u8_static", src, n, H0, W0, row_stride, frame_stride, chan_order);
    if (rc) return rc;
    Geometry g;
    if ((rc = check_net_shape(h, n, net_h, net_w, &g))) return rc;
    const StaticLayout lay = static_layout(g);
    if (n > lay.max_frames) {
        set_error("cd_forward_u8_static: %d frames of %d rows exceed the %d frames the %zu-byte static "
                  "activation array holds at that height; use cd_forward_u8", n, net_h, lay.max_frames,
                  kStaticScratchBytes);
        return CD_ECAP;
    }
    if (h->device >= kMaxDevices || !g_static[h->device].ptr) {
        set_error("cd_forward_u8_static: no static activation array on device %d", h->device);
        return CD_EINVAL;
    }
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    StaticScratch& ss = g_static[h->device];
    if ((rc = ensure_tables(h->rt, H0, W0, net_h, net_w, s))) return rc;
    {
        // The array keeps the previous call's contents.  Another input height, or another element
        // size over the same [rows][cols][48] cells (bf16 handles store 2-byte cells, fp32 and split
        // bf16 handles 4-byte ones), puts the zero borders elsewhere: clear it, and finish before a
        // later call on another stream could overtake.
        const int es = h->dtype == CD_BF16 ? 2 : 4;
        std::lock_guard<std::mutex> lock(g_static_mutex);
        if (ss.ws_h != g.h || ss.ws_es != es) {
            CD_HIP(hipMemsetAsync(ss.ptr, 0, kStaticScratchBytes, s));
            CD_HIP(hipStreamSynchronize(s));
            ss.ws_h = g.h;
            ss.ws_es = es;
        }
    }
    const Workspace ws{reinterpret_cast<float*>(ss.ptr), reinterpret_cast<float*>(ss.ptr + lay.act2_off)};
    return forward_u8_on(h, ws, h->rt, src, n, H0, W0, row_stride, frame_stride, chan_order, g, logits, maxv, argmax, s,
                         false);
}

int cd_forward_u8_range(cd_handle* h, const uint8_t* src, int64_t n, int batch, int H0, int W0, int64_t row_stride,
                        int64_t frame_stride, int chan_order, int net_h, int net_w, float* logits, float* maxv,
                        int64_t* argmax, void* stream) {
    if (!h || !logits) { set_error("cd_forward_u8_range: null argument"); return CD_EINVAL; }
    if (n < 1 || batch < 1) { set_error("cd_forward_u8_range: %lld frames in batches of %d", (long long)n, batch); return CD_EINVAL; }
    const int first = (int)std::min<int64_t>(n, batch);
    int rc = check_src("cd_forward_u8_range", src, first, H0, W0, row_stride, frame_stride, chan_order);
    if (rc) return rc;
    Geometry g;
    if ((rc = check_net_shape(h, first, net_h, net_w, &g))) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nb = (n + batch - 1) / batch;
    if (nb == 1) {       // one batch: exactly cd_forward_u8
        if ((rc = ensure_tables(h->rt, H0, W0, net_h, net_w, s))) return rc;
        if ((rc = ensure_workspace(h, 0, first, g, s))) return rc;
        return forward_u8_on(h, h->ws[0], h->rt, src, first, H0, W0, row_stride, frame_stride, chan_order, g, logits,
                             maxv, argmax, s, true);
    }
    // A geometry that needs a real resize goes through a stand-alone resize pass one batch ahead
    // (memory-bound) beside the classification of resized frames (matrix-bound); a point-sampling
    // geometry, and frames that already have the network's size, go straight into the fused block 1.
    // Split bf16: every geometry goes through the resize pass, which writes the first block's operand
    // cells, and blocks 1 + 2 run as one kernel whose block-1 activations stay in LDS.
    const bool fused12 = h->dtype == CD_BF16_SPLIT && h->opt[CD_OPT_FUSE_BLOCKS12] &&
                         resize_cells_supported(src, W0, row_stride, frame_stride, net_w);
    const bool two_pass = fused12 || (!(H0 == net_h && W0 == net_w) && !resize_is_point_sample(H0, W0, net_h, net_w));
    if ((rc = ensure_range_state(h))) return rc;
    const ResizeTables* rt = &h->rt;
    if (fused12) {
        if ((rc = ensure_tables(h->rt_pre, H0, W0, net_h, net_w, s))) return rc;
        const size_t slot = (size_t)(batch + 1) * cell_frame_bytes(net_h);
        if (slot > h->cells_bytes || h->cells_h != net_h) {
            CD_HIP(hipStreamSynchronize(s));
            CD_HIP(hipStreamSynchronize(h->side_cls));
            CD_HIP(hipStreamSynchronize(h->side_pre));
            if (slot > h->cells_bytes) {
                if (h->cells) CD_HIP(hipFree(h->cells));
                h->cells = nullptr; h->cells_bytes = 0;
                CD_HIP(hipMalloc(reinterpret_cast<void**>(&h->cells), 3 * slot));
                h->cells_bytes = slot;
            }
            // the margins of the cell frames are zero and never written; another height moves them
            CD_HIP(hipMemsetAsync(h->cells, 0, 3 * h->cells_bytes, s));
            CD_HIP(hipStreamSynchronize(s));
            h->cells_h = net_h;
        }
    } else if (two_pass) {
        if ((rc = ensure_tables(h->rt_pre, H0, W0, net_h, net_w, s))) return rc;
        if ((rc = ensure_tables(h->rt_id, net_h, net_w, net_h, net_w, s))) return rc;
        rt = &h->rt_id;
        const size_t slot = (size_t)batch * net_h * net_w * 3;
        if (slot > h->small_bytes) {
            CD_HIP(hipStreamSynchronize(s));
            CD_HIP(hipStreamSynchronize(h->side_cls));
            CD_HIP(hipStreamSynchronize(h->side_pre));
            if (h->small) CD_HIP(hipFree(h->small));
            h->small = nullptr; h->small_bytes = 0;
            CD_HIP(hipMalloc(reinterpret_cast<void**>(&h->small), 3 * slot + 16));   // + the 16-byte row-tail slack
            h->small_bytes = slot;
        }
    } else if ((rc = ensure_tables(h->rt, H0, W0, net_h, net_w, s))) return rc;
    if ((rc = ensure_workspace(h, 0, batch, g, s))) return rc;
    if ((rc = ensure_workspace(h, 1, batch, g, s))) return rc;

    // fork: the side streams start after everything already queued on the caller's stream
    CD_HIP(hipEventRecord(h->ev_fork, s));
    CD_HIP(hipStreamWaitEvent(h->side_cls, h->ev_fork, 0));
    if (two_pass) CD_HIP(hipStreamWaitEvent(h->side_pre, h->ev_fork, 0));
    const int64_t small_row = (int64_t)net_w * 3, small_frame = small_row * net_h;
    const int mode = fused12 ? 0 : h->opt[CD_OPT_RANGE_BY_STAGE];
    const bool one_stream = h->opt[CD_OPT_RANGE_ONE_STREAM] != 0;
    // Two classification streams at every size.  (While fp32's block 1 ran on the fp32 matrix instructions and the
    // resize pass held LDS, 1080p fp32 was faster with one: 402 k against 363 k frames/s.  With block 1 on the bf16
    // pipe and the LDS-free resize it is the other way round, 523 k against 504 k at batch 256; that remains true
    // with CD_OPT_F32_BLOCK1_FP32_MFMA only for small frames, so the old rule is kept for that setting.)
    const bool large_frames = (int64_t)H0 * W0 * 3 > (2 << 20);
    const bool one_cls_rule = two_pass && h->dtype == CD_F32 && h->opt[CD_OPT_F32_BLOCK1_FP32_MFMA] && large_frames;
#ifdef CD_EXP_ENV
    const bool two_cls = std::getenv("CUTDET_X_ONE_CLS") ? false : std::getenv("CUTDET_X_TWO_CLS") ? true : !one_cls_rule;
#else
    const bool two_cls = !one_cls_rule;
#endif
    // split bf16 beside a resize pass that saturates HBM: block 2 asks for its weights 8 steps ahead instead
    // of 2 (interleaved A/B at 1080p: +2.9 %; at 720p and 320x180, where nothing streams beside it, -0.5 %)
#ifdef CD_EXP_NO_FAR      // measurement build
    const bool far_weights = false;
#else
    const bool far_weights = two_pass && !one_stream && large_frames;
#endif
    hipStream_t pre = one_stream ? s : h->side_pre;
    for (int64_t i = 0; i < nb; ++i) {
        const int64_t a = i * batch;
        const int k = (int)std::min<int64_t>(batch, n - a);
        const int par = (int)(i & 1), slot = (int)(i % 3);
        // mode 0: the whole batch on one of two alternating streams.  mode 1: block 1 of every batch on the
        // side stream, blocks 2 and 3 on the caller's stream, handed over through the two workspaces.
        hipStream_t st1 = one_stream ? s : mode == 1 ? h->side_cls : (par && two_cls ? h->side_cls : s);
        hipStream_t st2 = one_stream || mode == 1 ? s : st1;
        const Workspace& ws = h->ws[par];
        const uint8_t* in = src + a * frame_stride;
        int fH = H0, fW = W0;
        int64_t frs = row_stride, ffs = frame_stride;
        if (fused12) {
            uint8_t* dst = h->cells + (size_t)slot * h->cells_bytes;
            if (i >= 3) CD_HIP(hipStreamWaitEvent(pre, h->ev_free[slot], 0));   // batch i-3 is done with the slot
            {
                StageTimer t(h, CD_STAGE_PREPROCESS, pre);
                if ((rc = launch_resize_cells(h, in, k, H0, W0, row_stride, frame_stride, chan_order, dst, net_h, net_w,
                                              pre))) return rc;
            }
            CD_HIP(hipEventRecord(h->ev_ready[slot], pre));
            CD_HIP(hipStreamWaitEvent(st1, h->ev_ready[slot], 0));
            {   // blocks 1 + 2 in one kernel, timed as block 2
                StageTimer t(h, CD_STAGE_CONV2, st1);
                if ((rc = launch_conv12_split(h, ws, dst, k, g, st1))) return rc;
            }
            CD_HIP(hipEventRecord(h->ev_free[slot], st1));
            StageTimer t(h, CD_STAGE_CONV3_HEAD, st1);
            if ((rc = launch_conv3_head(h, ws, k, g, logits + a * kOut, maxv ? maxv + a : nullptr,
                                        argmax ? argmax + a : nullptr, st1))) return rc;
            continue;
        }
        if (two_pass) {
            uint8_t* dst = h->small + (size_t)slot * h->small_bytes;
            if (i >= 3) CD_HIP(hipStreamWaitEvent(pre, h->ev_free[slot], 0));   // batch i-3 is done with the slot
            CD_SKIP_IF(1) {
                StageTimer t(h, CD_STAGE_PREPROCESS, pre);
                if ((rc = launch_resize_u8(h, in, k, H0, W0, row_stride, frame_stride, dst, net_h, net_w, pre))) return rc;
            }
            CD_HIP(hipEventRecord(h->ev_ready[slot], pre));
            CD_HIP(hipStreamWaitEvent(st1, h->ev_ready[slot], 0));
            in = dst; fH = net_h; fW = net_w; frs = small_row; ffs = small_frame;
        }
        if (mode == 1 && i >= 2) CD_HIP(hipStreamWaitEvent(st1, h->ev_a1free[par], 0));   // block 2 of batch i-2 has read act1
        CD_SKIP_IF(2) {
            StageTimer t(h, CD_STAGE_CONV1, st1);
            if ((rc = launch_conv1_u8(h, ws, *rt, in, k, fH, fW, frs, ffs, chan_order, g, st1))) return rc;
        }
        if (two_pass) CD_HIP(hipEventRecord(h->ev_free[slot], st1));
        if (mode == 1) {
            CD_HIP(hipEventRecord(h->ev_a1ready[par], st1));
            CD_HIP(hipStreamWaitEvent(st2, h->ev_a1ready[par], 0));
        }
        CD_SKIP_IF(4) { StageTimer t(h, CD_STAGE_CONV2, st2); if ((rc = launch_conv2(h, ws, k, g, st2, far_weights))) return rc; }
        if (mode == 1) CD_HIP(hipEventRecord(h->ev_a1free[par], st2));
        StageTimer t(h, CD_STAGE_CONV3_HEAD, st2);
        CD_SKIP_IF(8)
        if ((rc = launch_conv3_head(h, ws, k, g, logits + a * kOut, maxv ? maxv + a : nullptr,
                                    argmax ? argmax + a : nullptr, st2))) return rc;
    }
    // join: everything this call queued anywhere is ordered before what the caller queues next
    CD_HIP(hipEventRecord(h->ev_join, h->side_cls));
    CD_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
    return CD_OK;
}

int cd_set_option(cd_handle* h, int option, int value) {
    if (!h || option < 0 || option >= CD_OPT_COUNT || (value != 0 && value != 1)) {
        set_error("cd_set_option: option %d, value %d", option, value);
        return CD_EINVAL;
    }
    h->opt[option] = value;
    return CD_OK;
}

int cd_rle(cd_handle* h, const int64_t* cls, const float* maxv, int64_t n, int64_t frame_offset, int64_t* end,
           int64_t* type, int64_t* runlen, int64_t* start, float* mean, double* sum, int64_t cap,
           int64_t* n_seg, void* stream) {
    if (!h || !cls || !maxv || !end || !type || !runlen || !start || !mean || !n_seg) {
        set_error("cd_rle: null argument");
        return CD_EINVAL;
    }
    if (n < 1) { set_error("cd_rle: no frames (index out of bounds)"); return CD_EINVAL; }
    if (cap < 1) { set_error("cd_rle: capacity %lld", (long long)cap); return CD_ECAP; }
    DeviceGuard guard(h->device);
    StageTimer t(h, CD_STAGE_RLE, static_cast<hipStream_t>(stream));
    return launch_rle(h, cls, maxv, n, frame_offset, end, type, runlen, start, mean, sum, cap, n_seg,
                      static_cast<hipStream_t>(stream));
}

int cd_max_argmax(cd_handle* h, const float* scores, int64_t n, int n_classes, float* maxv, int64_t* argmax,
                  void* stream) {
    if (!h || !scores || !maxv || !argmax || n < 1 || n_classes < 1) {
        set_error("cd_max_argmax: null argument or empty input");
        return CD_EINVAL;
    }
    DeviceGuard guard(h->device);
    return launch_max_argmax(scores, n, n_classes, maxv, argmax, static_cast<hipStream_t>(stream));
}

int cd_eval_accumulate(cd_handle* h, const float* logits, const int64_t* labels, int64_t n, int n_classes,
                       int64_t* correct, int64_t* total, double* loss_sum, void* stream) {
    if (!h || !logits || !labels || !correct || !total || !loss_sum || n < 1 || n_classes < 1 || n_classes > 8) {
        set_error("cd_eval_accumulate: null argument, empty batch or more than 8 classes");
        return CD_EINVAL;
    }
    DeviceGuard guard(h->device);
    return launch_eval_accumulate(logits, labels, n, n_classes, correct, total, loss_sum,
                                  static_cast<hipStream_t>(stream));
}

#ifdef CD_DIAG
// diagnostic builds only: copy bytes out of the block-1 activation workspace
int cd_debug_peek_act1(cd_handle* h, size_t byte_off, size_t nbytes, void* host_dst) {
    if (!h || byte_off + nbytes > h->act1_bytes[0]) return CD_EINVAL;
    CD_HIP(hipDeviceSynchronize());
    CD_HIP(hipMemcpy(host_dst, reinterpret_cast<char*>(h->ws[0].act1) + byte_off, nbytes, hipMemcpyDeviceToHost));
    return CD_OK;
}
#endif

#ifdef CD_DIAG
int cd_debug_peek_act2(cd_handle* h, size_t byte_off, size_t nbytes, void* host_dst) {
    if (!h || byte_off + nbytes > h->act2_bytes[0]) return CD_EINVAL;
    CD_HIP(hipDeviceSynchronize());
    CD_HIP(hipMemcpy(host_dst, reinterpret_cast<char*>(h->ws[0].act2) + byte_off, nbytes, hipMemcpyDeviceToHost));
    return CD_OK;
}
#endif

int cd_timing_enable(cd_handle* h, int on) {
    if (!h) { set_error("cd_timing_enable: null handle"); return CD_EINVAL; }
    // 0 off, 1 every stage, otherwise bit (stage + 1) selects a stage
    h->timing = on == 1 ? ~0u : (unsigned)on >> 1;
    if (!on) for (int st = 0; st < CD_STAGE_COUNT; ++st) h->ev_used[st] = 0;
    return CD_OK;
}

int cd_timing_interval(cd_handle* h, int every) {
    if (!h || every < 1) { set_error("cd_timing_interval: bad argument"); return CD_EINVAL; }
    h->timing_every = every;
    for (unsigned& c : h->timing_seen) c = 0;
    return CD_OK;
}

int cd_timing_read(cd_handle* h, int stage, int64_t* launches, double* total_ms) {
    if (!h || !launches || !total_ms || stage < 0 || stage >= CD_STAGE_COUNT) {
        set_error("cd_timing_read: bad argument");
        return CD_EINVAL;
    }
    DeviceGuard guard(h->device);
    double ms = 0.0;
    for (size_t k = 0; k < h->ev_used[stage]; ++k) {
        CD_HIP(hipEventSynchronize(h->ev_stop[stage][k]));
        float one = 0.f;
        CD_HIP(hipEventElapsedTime(&one, h->ev_start[stage][k], h->ev_stop[stage][k]));
        ms += one;
    }
    *launches = (int64_t)h->ev_used[stage];
    *total_ms = ms;
    h->ev_used[stage] = 0;
    return CD_OK;
}

int cd_synth_frames(cd_handle* h, uint8_t* dst, int n, int H0, int W0, const uint8_t* classes, uint32_t seed,
                    int64_t first_index, void* stream) {
    if (!h || !dst || !classes || n < 1 || n > 65535 || H0 < 1 || W0 < 1) {
        set_error("cd_synth_frames: bad argument (1..65535 frames per call)");
        return CD_EINVAL;
    }
    DeviceGuard guard(h->device);
    return launch_synth(dst, n, H0, W0, classes, seed, first_index, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// Micro-benchmark: how fast can a CU pull every 5th row of 1280x720x3 u8 frames, with the access
// patterns the fused resize could use?  (Diagnostic tool, not part of the library.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t __attribute__((aligned(1))) u32u;
constexpr int W0 = 1280, H0 = 720, ROWB = W0 * 3, FRAME = ROWB * H0;

// workgroup b -> frame (XCD-aware) and chunk of 11 sampled rows (9 new + 2 halo), like conv1
__device__ __forceinline__ void locate(int b, int& frame, int& chunk) {
    const int xcd = b & 7, i = b >> 3;
    frame = xcd + 8 * (i / 16);
    chunk = i % 16;
}

template <int MODE, int ROWS>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, int burst) {
    int frame, chunk;
    locate(blockIdx.x, frame, chunk);
    const int tid = threadIdx.x;
    const uint8_t* f = src + (int64_t)frame * FRAME;
    uint32_t acc = 0;
    if (MODE == 0) {            // per-pixel unaligned dword gather, 15-byte stride
        const int xo = min((5 * max(tid - 1, 0) + 2) * 3, ROWB - 4);
        uint32_t v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int gy = min(max(9 * chunk - 1 + r, 0), 143);
            v[r] = *reinterpret_cast<const u32u*>(f + (int64_t)(5 * gy + 2) * ROWB + xo);
            if (burst == 0) acc += v[r];     // burst==0: consume each load before the next is issued
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc += v[r];
    } else if (MODE == 1) {     // coalesced 16 B per lane, 240 lanes cover a row
        uint4 v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int gy = min(max(9 * chunk - 1 + r, 0), 143);
            v[r] = tid < 240 ? *reinterpret_cast<const uint4*>(f + (int64_t)(5 * gy + 2) * ROWB + tid * 16) : uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
    } else {                    // one aligned dword per 128-byte line (30 lanes per row): line fetch only
        const int r = tid / 30, ln = tid % 30;
        if (r < ROWS) {
            const int gy = min(max(9 * chunk - 1 + r, 0), 143);
            acc = *reinterpret_cast<const uint32_t*>(f + (int64_t)(5 * gy + 2) * ROWB + ln * 128);
        }
        if (tid + 256 < ROWS * 30) {
            const int r2 = (tid + 256) / 30, l2 = (tid + 256) % 30;
            const int gy = min(max(9 * chunk - 1 + r2, 0), 143);
            acc += *reinterpret_cast<const uint32_t*>(f + (int64_t)(5 * gy + 2) * ROWB + l2 * 128);
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + tid] = acc;
}

int main() {
    const int n = 256;
    uint8_t* src; uint32_t* out;
    CK(hipMalloc(&src, (size_t)n * FRAME));
    CK(hipMemset(src, 1, (size_t)n * FRAME));
    CK(hipMalloc(&out, (size_t)n * 16 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int burst) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(n * 16), dim3(256), 0, 0, src, out, burst);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(n * 16), dim3(256), 0, 0, src, out, burst);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)n * 16 * 11 * ROWB;
        printf("%-44s %8.1f us/launch  %6.2f TB/s of touched lines\n", name, ms * 1000 / 20, bytes / (ms / 20 * 1e-3) / 1e12);
    };
    run("gather dword/pixel, 11 rows in flight", k<0, 11>, 1);
    run("gather dword/pixel, one row at a time", k<0, 11>, 0);
    run("coalesced 16 B/lane, 11 rows in flight", k<1, 11>, 1);
    run("one dword per 128 B line", k<2, 11>, 1);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}

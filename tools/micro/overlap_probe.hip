// Do a matrix-pipe-bound kernel and an HBM-bound kernel overlap when they are launched on two streams?
// A: persistent single-wave workgroups (one per SIMD) issuing back-to-back v_mfma_f32_16x16x32_bf16, with a
//    chosen register and LDS footprint.  B: a read stream over a 2 GiB buffer, 256-thread workgroups, a chosen
//    LDS footprint.  Prints A alone, B alone, and both together (two streams, same start).
//   hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int REGS>
__global__ __launch_bounds__(64) void mfma_kernel(float* out, int iters) {
    extern __shared__ float lds[];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)i; }
    if (REGS == 320) asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a63, 0" ::: "v255", "a63");   // forces the allocation up: 256 + 64
    if (REGS == 200) asm volatile("v_mov_b32 v190, 0" ::: "v190");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a8), "v"(b8));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) { out[0] = s; lds[threadIdx.x] = s; }
}

__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ p, size_t n, uint4* __restrict__ sink) {
    extern __shared__ float lds[];
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) { sink[0] = acc; lds[threadIdx.x] = 1.f; }
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    uint4 *buf, *sink; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(buf, 1, bytes));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int iters = 6000;                     // 48 000 MFMAs x 8 ns = ~380 us
    auto run = [&](const char* name, auto kern, int lds_a, int grid_b, int lds_b) {
        auto A = [&](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(1024), dim3(64), lds_a, s, out, iters); };
        auto B = [&](hipStream_t s) { hipLaunchKernelGGL(read_kernel, dim3(grid_b), dim3(256), lds_b, s, buf, n, sink); };
        float ta, tb, tab;
        A(sa); B(sb); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sa)); A(sa); CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ta, e0, e1));
        CK(hipEventRecord(e0, sb)); B(sb); CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&tb, e0, e1));
        CK(hipDeviceSynchronize());
        // together: B waits for A's start event so both begin at the same time
        CK(hipEventRecord(e0, sa)); CK(hipStreamWaitEvent(sb, e0, 0));
        A(sa); B(sb);
        CK(hipEventRecord(e1, sa)); CK(hipEventRecord(e2, sb));
        CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(e2));
        float t1, t2; CK(hipEventElapsedTime(&t1, e0, e1)); CK(hipEventElapsedTime(&t2, e0, e2));
        tab = t1 > t2 ? t1 : t2;
        printf("%-58s A %6.1f us  B %6.1f us (%.2f TB/s)  together %6.1f us (A done %6.1f, B done %6.1f)  sum %6.1f\n", name,
               ta * 1e3, tb * 1e3, bytes / tb / 1e9, tab * 1e3, t1 * 1e3, t2 * 1e3, (ta + tb) * 1e3);
    };
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel<320>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel<200>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    run("A: 64 regs, no LDS      B: 2048 WGs, no LDS", mfma_kernel<64>, 0, 2048, 0);
    run("A: 320 regs, 27 KB LDS  B: 2048 WGs, no LDS", mfma_kernel<320>, 27 * 1024, 2048, 0);
    run("A: 320 regs, 27 KB LDS  B: 2048 WGs, 11.5 KB LDS", mfma_kernel<320>, 27 * 1024, 2048, 11776);
    run("A: 320 regs, 27 KB LDS  B: 73728 WGs, 11.5 KB LDS", mfma_kernel<320>, 27 * 1024, 73728, 11776);
    run("A: 200 regs, 27 KB LDS  B: 2048 WGs, 11.5 KB LDS", mfma_kernel<200>, 27 * 1024, 2048, 11776);
    run("A: 64 regs, no LDS      B: 73728 WGs, 11.5 KB LDS", mfma_kernel<64>, 0, 73728, 11776);
    return 0;
}

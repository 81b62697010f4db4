// Micro-benchmark: do ordinary VALU instructions overlap with fp32 MFMAs on one SIMD?
// One wave per SIMD issues {1 MFMA + K independent v_fma} per iteration; a second experiment puts the
// VALU work in a second wave on the same SIMD.  (Diagnostic tool, not part of the library.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(64) void same_wave(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x, b = 1.0f, x[8];
    for (int i = 0; i < 8; ++i) x[i] = i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[k & 7]) : "v"(b));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + x[i];
    if (s == 12345.f) out[0] = s;
}

// waves 0..3 of a 512-thread block: MFMA only; waves 4..7 (second wave of each SIMD): VALU only
template <int MODE>
__global__ __launch_bounds__(512) void two_waves(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x, b = 1.0f, x[8];
    for (int i = 0; i < 8; ++i) x[i] = i + threadIdx.x;
    if (wave < 4) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
    } else if (MODE == 1) {
        if (MODE == 1) __builtin_amdgcn_s_setprio(3);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[u & 7]) : "v"(b));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + x[i];
    if (s == 12345.f) out[0] = s;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    auto run = [&](const char* name, auto kern, dim3 grid, dim3 block, double mfma_per_wave) {
        hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %8.3f ms   %6.1f ns per MFMA\n", name, ms, ms * 1e6 / mfma_per_wave);
    };
    const double m = (double)iters * 8;
    run("1 wave/SIMD: MFMA only", same_wave<0>, dim3(1024), dim3(64), m);
    run("1 wave/SIMD: MFMA + 2 v_fma", same_wave<2>, dim3(1024), dim3(64), m);
    run("1 wave/SIMD: MFMA + 4 v_fma", same_wave<4>, dim3(1024), dim3(64), m);
    run("1 wave/SIMD: MFMA + 6 v_fma", same_wave<6>, dim3(1024), dim3(64), m);
    run("1 wave/SIMD: MFMA + 8 v_fma", same_wave<8>, dim3(1024), dim3(64), m);
    run("2 waves/SIMD: MFMA wave + idle wave", two_waves<0>, dim3(256), dim3(512), m);
    run("2 waves/SIMD: MFMA wave + VALU wave (8 fma/MFMA)", two_waves<1>, dim3(256), dim3(512), m);
    hipDeviceSynchronize();
    return 0;
}

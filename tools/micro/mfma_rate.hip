// Micro-benchmark: issue rate of the MFMA shapes the kernels use, one wave per SIMD (512-register
// waves so that exactly one fits).  (Diagnostic tool, not part of the library.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float a = threadIdx.x * 0.001f, b = 1.0f;
    s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {1, 2, 3, 4};
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // inline asm with VGPR accumulators: left to the builtins, the compiler permutes the
            // accumulators through AGPRs on every trip of the loop
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a4), "v"(b4));
            if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a8), "v"(b8));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    auto run = [&](const char* name, auto kern, double flop) {
        // 160 KB of dynamic LDS per workgroup would also pin one wave per CU; here 256 x 4 waves with
        // 64 KB each leave exactly one workgroup of 4 waves... simpler: one 256-thread block per CU
        hipLaunchKernelGGL(kern, dim3(256), dim3(64), 160 * 1024 / 4, 0, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64), 160 * 1024 / 4, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-32s %7.2f ns per MFMA  (%5.1f cycles at 2.4 GHz)  %7.1f GFLOP/s per SIMD\n", name,
               ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) * 2.4, flop / (ms * 1e6 / (iters * 8.0)));
    };
    run("v_mfma_f32_16x16x4_f32", k<0>, 2048.0);
    run("v_mfma_f32_16x16x16_bf16", k<1>, 8192.0);
    run("v_mfma_f32_16x16x32_bf16", k<2>, 16384.0);
    hipDeviceSynchronize();
    return 0;
}

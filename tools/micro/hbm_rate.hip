// What HBM delivers on this MI355X: read-only, write-only and copy streams over buffers far larger than the
// caches (4 GiB), 16 bytes per lane, grid-stride.   hipcc --offload-arch=gfx950 -O3 hbm_rate.hip -o hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ p, size_t n, uint4* __restrict__ sink) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc;      // never true on the test data: keeps the loads
}
__global__ __launch_bounds__(256) void write_kernel(uint4* __restrict__ p, size_t n) {
    const uint4 v = {1u, 2u, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

template <typename F> static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    uint4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    for (int wgs_per_cu : {4, 8, 16, 32}) {
        const unsigned grid = 256u * wgs_per_cu;
        const double r = time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n, b); }, 10);
        const double w = time_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, b, n); }, 10);
        const double c = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n); }, 10);
        printf("%2d workgroups/CU: read %.2f TB/s   write %.2f TB/s   copy %.2f TB/s (read + written bytes)\n", wgs_per_cu,
               bytes / r / 1e9, bytes / w / 1e9, 2.0 * bytes / c / 1e9);
    }
    return 0;
}

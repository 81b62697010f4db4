// Where do the two waves of a 128-thread workgroup land?  Prints, for the first workgroups of a
// persistent-style launch (1024 workgroups, 40 KB of LDS each = 4 per CU), the (XCC, SE, CU, SIMD,
// wave slot) of wave 0 and wave 1.   hipcc --offload-arch=gfx950 -O2 hwid_probe.hip -o hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(128, 2) void probe(unsigned* out, int spin) {
    __shared__ unsigned char lds[40576];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = (unsigned char)hw;
    __syncthreads();
    // stay resident long enough for every workgroup of the launch to be placed
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw + lds[5] * 0;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main() {
    const int n = 1024;
    unsigned* d;
    hipMalloc(&d, n * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(n), dim3(128), 0, 0, d, 20000);   // 200 us at 100 MHz
    std::vector<unsigned> h(n * 4);
    hipMemcpy(h.data(), d, n * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    int same_simd = 0, next_simd = 0, other = 0, same_slot = 0;
    for (int b = 0; b < n; ++b) {
        unsigned a = h[b * 4], c = h[b * 4 + 2];
        int s0 = (a >> 4) & 3, s1 = (c >> 4) & 3, w0 = a & 15, w1 = c & 15;
        if (s0 == s1) ++same_simd; else if (((s0 + 1) & 3) == s1) ++next_simd; else ++other;
        if (w0 == w1) ++same_slot;
        if (b < 24 || (b % 8 == 0 && b < 200))
            printf("wg %4d  xcc %u cu %2u se %u | wave0 simd %d slot %d | wave1 simd %d slot %d | raw %08x %08x\n", b, h[b * 4 + 1] & 15,
                   (a >> 8) & 15, (a >> 13) & 7, s0, w0, s1, w1, a, c);
    }
    printf("waves of a workgroup: same SIMD %d, next SIMD %d, other %d; same slot %d of %d\n", same_simd, next_simd, other, same_slot, n);
    return 0;
}

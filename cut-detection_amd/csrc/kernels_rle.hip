// Run-length encoding of the per-frame class stream (reference frameID/segmentation.py:39-60).
//
// A frame i ends a segment when it is the last frame or cls[i+1] != cls[i].  Three passes:
//   count    per-block number of segment ends (wave ballots + popcounts)
//   offsets  one workgroup turns the block counts into exclusive offsets and the total
//   scatter  every end frame learns its segment index from a wave-level ballot scan
//            (popcount of the lanes below it) on top of the running block offset
//   finish   start = previous end + 1, run length, and the per-segment mean of the winning
//            logit, one wave per segment.  The reference forms that mean with torch.mean on a
//            float32 CPU slice (segmentation.py:54-59) and later picks the orphan with the smallest
//            mean (:104-107), so the fp32 rounding of the mean is part of the result: for segments
//            below 32768 frames the wave adds the values in exactly the order of torch's CPU sum
//            kernel (see torch_order_sum) and the mean is that reference's bit for bit.  From 32768
//            elements on torch splits the reduction over its threads, so the reference's own value
//            depends on the machine; there the mean is an fp64 sum rounded once, taken from
//            per-block fp64 sums of the count pass plus the segment's ragged ends, which also keeps
//            a single hours-long segment off one wavefront's back.
#include <algorithm>

#include "cd_internal.h"

namespace cd {

constexpr int kRleThreads = 256;
constexpr int kRleItems = 16;                       // frames per thread
constexpr int kRleBlock = kRleThreads * kRleItems;  // frames per workgroup

__device__ __forceinline__ bool is_end(const int64_t* __restrict__ cls, int64_t i, int64_t n) {
    return i < n && (i == n - 1 || cls[i + 1] != cls[i]);
}

__global__ __launch_bounds__(kRleThreads) void rle_count_kernel(const int64_t* __restrict__ cls,
                                                               const float* __restrict__ maxv, int64_t n,
                                                               int64_t* __restrict__ block_counts,
                                                               double* __restrict__ block_sums) {
    __shared__ int wave_counts[kRleThreads / 64];
    __shared__ double wave_sums[kRleThreads / 64];
    const int64_t b0 = (int64_t)blockIdx.x * kRleBlock;
    int cnt = 0;
    double acc = 0.0;           // fp64 sum of this block's maxima, in a fixed order
    for (int it = 0; it < kRleItems; ++it) {
        const int64_t i = b0 + it * kRleThreads + threadIdx.x;
        unsigned long long m = __ballot(is_end(cls, i, n));
        cnt += __popcll(m);
        if (i < n) acc += (double)maxv[i];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) { wave_counts[threadIdx.x >> 6] = cnt; wave_sums[threadIdx.x >> 6] = acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        double sum = 0.0;
        for (int w = 0; w < kRleThreads / 64; ++w) { t += wave_counts[w]; sum += wave_sums[w]; }
        block_counts[blockIdx.x] = t;
        block_sums[blockIdx.x] = sum;
    }
}

// in place: counts -> exclusive offsets; *n_seg = total
__global__ __launch_bounds__(256) void rle_offsets_kernel(int64_t* __restrict__ blocks, int nblocks,
                                                          int64_t* __restrict__ n_seg) {
    __shared__ int64_t part[256];
    const int tid = threadIdx.x;
    const int per = (nblocks + 255) / 256;
    const int lo = min(tid * per, nblocks), hi = min(lo + per, nblocks);
    int64_t s = 0;
    for (int k = lo; k < hi; ++k) s += blocks[k];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int k = 0; k < 256; ++k) { int64_t v = part[k]; part[k] = run; run += v; }
        *n_seg = run;
    }
    __syncthreads();
    int64_t run = part[tid];
    for (int k = lo; k < hi; ++k) { int64_t v = blocks[k]; blocks[k] = run; run += v; }
}

__global__ __launch_bounds__(kRleThreads) void rle_scatter_kernel(
    const int64_t* __restrict__ cls, int64_t n, int64_t frame_offset, const int64_t* __restrict__ block_offsets,
    int64_t* __restrict__ end, int64_t* __restrict__ type, int64_t cap) {
    __shared__ int wave_counts[kRleThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * kRleBlock;
    int64_t run = block_offsets[blockIdx.x];
    for (int it = 0; it < kRleItems; ++it) {
        const int64_t i = b0 + it * kRleThreads + threadIdx.x;
        const bool e = is_end(cls, i, n);
        const unsigned long long m = __ballot(e);
        if (lane == 0) wave_counts[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < kRleThreads / 64; ++w) {
            int c = wave_counts[w];
            if (w < wave) before += c;
            total += c;
        }
        if (e) {
            int64_t k = run + before + __popcll(m & ((1ull << lane) - 1ull));
            if (k < cap) { end[k] = i + frame_offset; type[k] = cls[i]; }
        }
        run += total;
        __syncthreads();
    }
}

// fp32 sum of x[0..n) (8 <= n < 32768 uses the vector path, n < 8 the scalar one) in the order of
// torch's CPU sum kernel for a contiguous float32 row, which torch.mean divides by n:
//   * the row is read as 8-float vectors; vector v goes to accumulator v % 4 (four interleaved
//     vector accumulators), i.e. 32 independent fp32 chains: chain (k, l) adds x[(4i + k) * 8 + l],
//     i = 0, 1, ...  Lane (k = lane >> 3, l = lane & 7) of the wave runs chain (k, l);
//   * within a chain, every 16 elements are summed from zero and added to a second-level
//     accumulator, every 16 of those to a third, and so on (cascade summation, 4 levels); the
//     levels are added together at the end, lowest first;
//   * vectors left over after the last group of four go to accumulator 0; the four accumulators
//     are added as ((a0 + a1) + a2) + a3; then, starting from zero, the scalar tail (n % 8
//     elements) and the eight lanes are added one after the other.
// Every lane of the wave must call this (shuffles); the result is the same in all lanes.
__device__ __forceinline__ float torch_order_sum(const float* __restrict__ x, int n, int lane) {
    if (n < 8) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const int q = n >> 2;
        for (int i = 0; i < q; ++i)
            for (int k = 0; k < 4; ++k) a[k] += x[4 * i + k];
        for (int i = 4 * q; i < n; ++i) a[0] += x[i];
        return ((a[0] + a[1]) + a[2]) + a[3];
    }
    const int vec = n >> 3, ilp = vec >> 2;
    const int k = (lane >> 3) & 3, l = lane & 7;          // lanes 32..63 repeat lanes 0..31
    const int clog = ilp <= 2 ? 1 : 32 - __clz(ilp - 1);
    const int power = max(4, clog >> 2), step = 1 << power, mask = step - 1;
    const float* p = x + k * 8 + l;                       // chain element i is p[32 * i]
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    while (i + step <= ilp) {
        for (int j = 0; j < step; ++j, ++i) a0 += p[32 * i];
        a1 += a0; a0 = 0.f;
        if (!(i & (mask << power))) {
            a2 += a1; a1 = 0.f;
            if (!(i & (mask << (2 * power)))) { a3 += a2; a2 = 0.f; }
        }
    }
    for (; i < ilp; ++i) a0 += p[32 * i];
    a0 += a1; a0 += a2; a0 += a3;
    if (k == 0)
        for (int v = 4 * ilp; v < vec; ++v) a0 += x[v * 8 + l];
    const float s1 = __shfl(a0, l + 8), s2 = __shfl(a0, l + 16), s3 = __shfl(a0, l + 24);
    const float lane_sum = ((a0 + s1) + s2) + s3;         // meaningful in lanes 0..7
    float total = 0.f;
    for (int t = vec * 8; t < n; ++t) total += x[t];
    for (int t = 0; t < 8; ++t) total += __shfl(lane_sum, t);
    return total;
}

constexpr int64_t kTorchSplit = 32768;     // torch's grain size: from here on its reduction is split over threads

__global__ __launch_bounds__(256) void rle_finish_kernel(
    const float* __restrict__ maxv, int64_t frame_offset, const int64_t* __restrict__ end,
    int64_t* __restrict__ runlen, int64_t* __restrict__ start, float* __restrict__ mean,
    double* __restrict__ sum, int64_t cap, const int64_t* __restrict__ n_seg,
    const double* __restrict__ block_sums) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t S = *n_seg < cap ? *n_seg : cap;
    for (int64_t k = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < S; k += waves) {
        const int64_t e = end[k];
        const int64_t b = k == 0 ? frame_offset : end[k - 1] + 1;
        const int64_t len = e - b + 1;
        const int64_t lo = b - frame_offset, hi = e - frame_offset + 1;     // positions [lo, hi) of the stream
        // fp64 sum: whole 4096-frame blocks inside the segment come from the count pass
        const int64_t fb = (lo + kRleBlock - 1) / kRleBlock, lb = hi / kRleBlock;
        double acc = 0.0;
        if (fb < lb) {
            for (int64_t i = lo + lane; i < fb * kRleBlock; i += 64) acc += (double)maxv[i];
            for (int64_t q = fb + lane; q < lb; q += 64) acc += block_sums[q];
            for (int64_t i = lb * kRleBlock + lane; i < hi; i += 64) acc += (double)maxv[i];
        } else {
            for (int64_t i = lo + lane; i < hi; i += 64) acc += (double)maxv[i];
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        float m;
        if (len < kTorchSplit) m = torch_order_sum(maxv + lo, (int)len, lane) / (float)len;
        else m = (float)(acc / (double)len);
        if (lane == 0) {
            start[k] = b;
            runlen[k] = len;
            mean[k] = m;
            if (sum) sum[k] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void max_argmax_kernel(const float* __restrict__ scores, int64_t n, int ncls,
                                                         float* __restrict__ maxv, int64_t* __restrict__ argmax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* r = scores + i * ncls;
    float best = r[0];
    int bi = 0;
    for (int o = 1; o < ncls; ++o) {
        float v = r[o];
        if (!(best != best) && (v > best || v != v)) { best = v; bi = o; }
    }
    maxv[i] = best;
    argmax[i] = bi;
}

// Per-class correct / total counts and the summed cross-entropy of one labelled batch
// (reference training_scripts/supervised_training.py:186-201).  One frame per lane; counts leave a
// wave as ballots + popcounts, the loss as an fp64 butterfly sum, one atomic per wave and quantity.
__global__ __launch_bounds__(256) void eval_accumulate_kernel(const float* __restrict__ logits,
                                                              const int64_t* __restrict__ labels, int64_t n, int ncls,
                                                              unsigned long long* __restrict__ correct,
                                                              unsigned long long* __restrict__ total,
                                                              double* __restrict__ loss_sum) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int label = -1, pred = -2;
    double loss = 0.0;
    if (i < n) {
        const int64_t l = labels[i];
        const float* r = logits + i * ncls;
        float best = r[0];
        int bi = 0;
        for (int o = 1; o < ncls; ++o) {
            const float v = r[o];
            if (!(best != best) && (v > best || v != v)) { best = v; bi = o; }
        }
        if (l >= 0 && l < ncls) {
            label = (int)l;
            pred = bi;
            // -log_softmax(r)[label] = log(sum exp(r - max)) + max - r[label]
            double se = 0.0;
            for (int o = 0; o < ncls; ++o) se += exp((double)r[o] - (double)best);
            loss = log(se) + (double)best - (double)r[label];
        }
    }
    for (int c = 0; c < ncls; ++c) {
        const unsigned long long mt = __ballot(label == c), mc = __ballot(label == c && pred == c);
        if (lane == 0) {
            if (mt) atomicAdd(total + c, (unsigned long long)__popcll(mt));
            if (mc) atomicAdd(correct + c, (unsigned long long)__popcll(mc));
        }
    }
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
    if (lane == 0 && loss != 0.0) atomicAdd(loss_sum, loss);
}

int launch_eval_accumulate(const float* logits, const int64_t* labels, int64_t n, int ncls, int64_t* correct,
                           int64_t* total, double* loss_sum, hipStream_t s) {
    hipLaunchKernelGGL(eval_accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, logits, labels, n,
                       ncls, reinterpret_cast<unsigned long long*>(correct),
                       reinterpret_cast<unsigned long long*>(total), loss_sum);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

int launch_max_argmax(const float* scores, int64_t n, int ncls, float* maxv, int64_t* argmax, hipStream_t s) {
    hipLaunchKernelGGL(max_argmax_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scores, n, ncls, maxv,
                       argmax);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

int launch_rle(const cd_handle* h, const int64_t* cls, const float* maxv, int64_t n, int64_t frame_offset,
               int64_t* end, int64_t* type, int64_t* runlen, int64_t* start, float* mean, double* sum,
               int64_t cap, int64_t* n_seg, hipStream_t s) {
    const int64_t nblocks = (n + kRleBlock - 1) / kRleBlock;
    if (nblocks > h->rle_blocks_cap) {
        set_error("cd_rle: %lld frames exceed the %lld-frame limit of one call", (long long)n,
                  (long long)h->rle_blocks_cap * kRleBlock);
        return CD_EINVAL;
    }
    hipLaunchKernelGGL(rle_count_kernel, dim3((unsigned)nblocks), dim3(kRleThreads), 0, s, cls, maxv, n, h->rle_blocks,
                       h->rle_sums);
    hipLaunchKernelGGL(rle_offsets_kernel, dim3(1), dim3(256), 0, s, h->rle_blocks, (int)nblocks, n_seg);
    hipLaunchKernelGGL(rle_scatter_kernel, dim3((unsigned)nblocks), dim3(kRleThreads), 0, s, cls, n, frame_offset,
                       h->rle_blocks, end, type, cap);
    const unsigned fin_blocks = (unsigned)std::min<int64_t>(1024, (std::min<int64_t>(n, cap) + 3) / 4);
    hipLaunchKernelGGL(rle_finish_kernel, dim3(fin_blocks), dim3(256), 0, s, maxv, frame_offset, end, runlen,
                       start, mean, sum, cap, n_seg, h->rle_sums);
    CD_HIP(hipGetLastError());
    return CD_OK;
}

}  // namespace cd

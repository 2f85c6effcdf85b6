// Confusion-matrix accumulation of the validation metrics:
//   k_confusion   UnsupervisedMetrics.update / update_cherry   (reference src/utils.py:222-232, 279-289)
// stats[pred][actual] += 1 for every pixel with 0 <= actual < n_classes and 0 <= pred < n_classes - the reference masks the
// predictions with n_classes too, so clusters n_classes..n_classes+extra-1 never reach the matrix (rows stay zero).
// Integer counts: per-block histogram in LDS (32-bit), then one 64-bit atomic per non-empty bin - the result does not
// depend on the order, bit-identical to torch.bincount.
#include "dg_common.h"

#define CONF_THREADS 256

template <bool LDSHIST>
__global__ __launch_bounds__(CONF_THREADS) void k_confusion(const long long* __restrict__ preds, const long long* __restrict__ target,
                                                            long long count, int ncls, int nrows, unsigned long long* __restrict__ stats) {
    extern __shared__ unsigned int hist[];            // [nrows][ncls]
    const int bins = nrows * ncls;
    if (LDSHIST) {
        for (int b = threadIdx.x; b < bins; b += CONF_THREADS) hist[b] = 0u;
        __syncthreads();
    }
    const long long stride = (long long)gridDim.x * CONF_THREADS;
    auto count_one = [&](long long a, long long p) {
        if (a >= 0 && a < ncls && p >= 0 && p < ncls) {
            const int b = (int)p * ncls + (int)a;
            if (LDSHIST) atomicAdd(&hist[b], 1u);
            else atomicAdd(&stats[b], 1ull);
        }
    };
    long long i = (long long)blockIdx.x * CONF_THREADS + threadIdx.x;
    for (; i + 3 * stride < count; i += 4 * stride) {        // eight independent loads in flight per thread
        long long a[4], p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = __builtin_nontemporal_load(target + i + u * stride); p[u] = __builtin_nontemporal_load(preds + i + u * stride); }
#pragma unroll
        for (int u = 0; u < 4; ++u) count_one(a[u], p[u]);
    }
    for (; i < count; i += stride) count_one(target[i], preds[i]);
    if (LDSHIST) {
        __syncthreads();
        for (int b = threadIdx.x; b < bins; b += CONF_THREADS) {
            const unsigned int c = hist[b];
            if (c) atomicAdd(&stats[b], (unsigned long long)c);
        }
    }
}

hipError_t dg_launch_confusion(const long long* preds, const long long* target, long long count, int ncls, int nrows,
                               unsigned long long* stats, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    const size_t smem = (size_t)nrows * ncls * 4;
    long long blocks = (count + CONF_THREADS * 8 - 1) / (CONF_THREADS * 8);      // >= 8 pixels per thread
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (smem <= 64 * 1024) {
        hipLaunchKernelGGL(k_confusion<true>, dim3((unsigned)blocks), dim3(CONF_THREADS), smem, s, preds, target, count, ncls, nrows, stats);
    } else {
        hipLaunchKernelGGL(k_confusion<false>, dim3((unsigned)blocks), dim3(CONF_THREADS), 0, s, preds, target, count, ncls, nrows, stats);
    }
    return hipGetLastError();
}

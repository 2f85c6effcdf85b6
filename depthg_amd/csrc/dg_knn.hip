// Row-wise top-k of a similarity matrix: the selection step of the offline k-nearest-neighbour
// search, reference src/precompute_knns.py:108-112 (`torch.topk(pairwise_sims, 30)[1]` on every row slice of
// `einsum("nf,mf->nm")`).  The contraction itself is a plain GEMM and stays a library call in the host mirror
// (depthg_amd/knn.py); this kernel reads every similarity row once from HBM and a few more times from L2:
//   4 x 8-bit radix passes over order-preserving keys find the k-th largest key and how many keys are above it,
//   one ordered compaction collects those plus the first ties in index order, a bitonic sort orders the k winners.
// Output order: value descending, ties by ascending index (torch.topk leaves the tie order unspecified).
// One block per row, 256 threads.
#include "dg_common.h"

#define TOPK_THREADS 256
#define TOPK_MAXK 64

__device__ __forceinline__ unsigned int topk_key(float x) {          // larger float <-> larger key
    const unsigned int u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(TOPK_THREADS) void k_topk_rows(const float* __restrict__ vals, long long cols, long long row_stride,
                                                            int k, long long* __restrict__ out_idx, float* __restrict__ out_val) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sel_prefix, sel_remaining;
    __shared__ unsigned int wave_cnt[2][TOPK_THREADS / 64];
    __shared__ unsigned int taken[2];                                    // winners above the threshold / ties taken so far
    __shared__ unsigned long long win[TOPK_MAXK];                        // (key << 32) | ~index  -> sorts value desc, index asc
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* row = vals + (size_t)blockIdx.x * row_stride;
    if (tid == 0) { sel_prefix = 0u; sel_remaining = (unsigned int)k; }
    // ---- radix select: the k-th largest key, most significant byte first
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0u;
        __syncthreads();
        const unsigned int prefix = sel_prefix;
        const unsigned int himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (long long i = tid; i < cols; i += TOPK_THREADS) {
            const unsigned int key = topk_key(row[i]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        if (tid == 0) {                                                    // walk the bins from the top
            unsigned int rem = sel_remaining, b = 255u;
            for (;; --b) {
                const unsigned int c = hist[b];
                if (c >= rem || b == 0u) break;
                rem -= c;
            }
            sel_prefix = prefix | (b << shift);
            sel_remaining = rem;                                           // rank of the wanted key inside bin b
        }
        __syncthreads();
    }
    const unsigned int T = sel_prefix;                                     // k-th largest key
    const unsigned int need_ties = sel_remaining;                          // how many keys == T belong to the top k
    const unsigned int above = (unsigned int)k - need_ties;                // keys > T: all of them are winners
    if (tid < 2) taken[tid] = 0u;
    __syncthreads();
    // ---- ordered compaction: keys > T in any order, keys == T in index order (lowest indices first)
    for (long long i0 = 0; i0 < cols; i0 += TOPK_THREADS) {
        const long long i = i0 + tid;
        unsigned int key = 0u;
        bool gt = false, eq = false;
        if (i < cols) { key = topk_key(row[i]); gt = key > T; eq = key == T; }
        const unsigned long long mg = __ballot(gt), me = __ballot(eq);
        if (lane == 0) { wave_cnt[0][wid] = (unsigned int)__popcll(mg); wave_cnt[1][wid] = (unsigned int)__popcll(me); }
        __syncthreads();
        unsigned int bg = taken[0], be = taken[1], tg = 0u, te = 0u;
        for (int w = 0; w < TOPK_THREADS / 64; ++w) {
            if (w < wid) { bg += wave_cnt[0][w]; be += wave_cnt[1][w]; }
            tg += wave_cnt[0][w]; te += wave_cnt[1][w];
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long entry = ((unsigned long long)key << 32) | (unsigned int)(~(unsigned int)i);
        if (gt) win[bg + (unsigned int)__popcll(mg & below)] = entry;
        if (eq) {
            const unsigned int pos = be + (unsigned int)__popcll(me & below);
            if (pos < need_ties) win[above + pos] = entry;
        }
        __syncthreads();
        if (tid == 0) { taken[0] += tg; taken[1] += te; }
        __syncthreads();
        if (taken[0] >= above && taken[1] >= need_ties) break;            // (uniform: every thread reads the same LDS words)
    }
    // ---- sort the k winners (descending entries) with one wave; k <= 64
    if (wid == 0) {
        unsigned long long e = lane < k ? win[lane] : 0ull;
        for (int kk = 2; kk <= 64; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                const unsigned long long o = __shfl_xor(e, j, 64);
                const bool up = (lane & kk) == 0;                          // descending overall
                const bool lower = (lane & j) == 0;
                const bool take_max = up == lower;
                e = take_max ? (e > o ? e : o) : (e < o ? e : o);
            }
        if (lane < k) {
            const unsigned int idx = ~(unsigned int)(e & 0xFFFFFFFFull);
            out_idx[(size_t)blockIdx.x * k + lane] = (long long)idx;
            if (out_val) out_val[(size_t)blockIdx.x * k + lane] = row[idx];
        }
    }
}

hipError_t dg_launch_topk_rows(const float* vals, long long rows, long long cols, long long row_stride, int k,
                               long long* out_idx, float* out_val, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_topk_rows, dim3((unsigned)rows), dim3(TOPK_THREADS), 0, s, vals, cols, row_stride, k, out_idx, out_val);
    return hipGetLastError();
}

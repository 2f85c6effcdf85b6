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

// ------------------------------------------------------------------------------------------
// The contraction of the nearest-neighbour search itself (reference src/precompute_knns.py:106-108,
// `pairwise_sims = einsum("nf,mf->nm", batch_feats, normed_feats)`): sims[i][j] = <q_i, x_j> in fp32 on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulation in k order - an fp32 dot product, no reduced-precision
// operands, so near-ties between neighbours fall as they do for an fp32 GEMM).  Measured on a 775 x 49,629 x 384 slice
// (scripts/knn_time.py): this kernel 75 TFLOP/s, the vendor GEMM (torch.matmul -> rocBLAS) 105 TFLOP/s; the whole cocostuff-sized
// table 52 against 45 ms.  It is the default engine all the same (depthg_amd/knn.py says why: a fixed k order -> the same table
// bits on every ROCm version; the 7 ms are paid once, offline).
// One block = 128 query rows x 128 candidate rows, four waves of 64 x 64 (2 x 2 accumulator tiles); K in chunks of 32 through
// two LDS buffers, rows of 36 floats ([row][k], as in memory: the staging is 16-byte loads -> 16-byte LDS stores), the next chunk
// register-staged during the MFMAs of the current one.  The MFMA pairs k-index h of a lane's operand with the other half
// wave's: a dot product does not care which two k share a step, so the lanes of half h take k = 16 h .. 16 h + 15 of the chunk -
// sixteen CONTIGUOUS floats of their row, four ds_read_b128 per operand tile and chunk (the first version read one float per
// MFMA operand and wrote the staged tile with 4-byte stores: 77 TFLOP/s).
#define SIMS_TM 128
#define SIMS_KC 32
#define SIMS_LD 36
__global__ __launch_bounds__(256) void k_sims_nt(const float* __restrict__ q, const float* __restrict__ x, long long rows_q, long long n,
                                                 int F, long long q_stride, long long x_stride, float* __restrict__ out, long long out_stride) {
    __shared__ __attribute__((aligned(16))) float As[2][SIMS_TM * SIMS_LD], Bs[2][SIMS_TM * SIMS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    // (the query tiles of one candidate tile are neighbours in the grid: the candidate rows come from HBM once per slice)
    const long long q0 = (long long)blockIdx.x * SIMS_TM, x0 = (long long)blockIdx.y * SIMS_TM;
    const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;                 // this wave's 64 x 64 corner of the block tile
    // staging: float4 piece i of a chunk = row i / 8, k-quad i % 8; four pieces per thread and operand
    f32x4 pa[4], pb[4];
    // 16-byte loads where every row starts on a 16-byte boundary and F is a multiple of 4; scalar loads otherwise (any stride)
    const bool vec = (F & 3) == 0 && ((q_stride | x_stride) & 3) == 0 && ((((uintptr_t)q) | ((uintptr_t)x)) & 15) == 0;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, row = i >> 3, kq = (i & 7) * 4 + k0;
            const long long qa = q0 + row, xb = x0 + row;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (vec && kq + 3 < F) {                 // (F a multiple of 4: the launcher's caller guarantees 16-byte aligned rows)
                if (qa < rows_q) va = *reinterpret_cast<const f32x4*>(q + qa * q_stride + kq);
                if (xb < n) vb = *reinterpret_cast<const f32x4*>(x + xb * x_stride + kq);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (kq + e < F && qa < rows_q) va[e] = q[qa * q_stride + kq + e];
                    if (kq + e < F && xb < n) vb[e] = x[xb * x_stride + kq + e];
                }
            }
            pa[u] = va; pb[u] = vb;
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, row = i >> 3, kq = (i & 7) * 4;
            *reinterpret_cast<f32x4*>(&As[b][row * SIMS_LD + kq]) = pa[u];
            *reinterpret_cast<f32x4*>(&Bs[b][row * SIMS_LD + kq]) = pb[u];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{};
    const int nchunk = (F + SIMS_KC - 1) / SIMS_KC;
    fetch(0);
    stash(0);
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) fetch((c + 1) * SIMS_KC);
        __syncthreads();                                                   // chunk c is in buffer c & 1; the other one is free
        const float* A = As[c & 1] + (wm + r) * SIMS_LD + 16 * h;
        const float* Bm = Bs[c & 1] + (wn + r) * SIMS_LD + 16 * h;
        f32x4 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a0[g] = *reinterpret_cast<const f32x4*>(A + 4 * g);
            a1[g] = *reinterpret_cast<const f32x4*>(A + 32 * SIMS_LD + 4 * g);
            b0[g] = *reinterpret_cast<const f32x4*>(Bm + 4 * g);
            b1[g] = *reinterpret_cast<const f32x4*>(Bm + 32 * SIMS_LD + 4 * g);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float av0 = a0[j >> 2][j & 3], av1 = a1[j >> 2][j & 3], bv0 = b0[j >> 2][j & 3], bv1 = b1[j >> 2][j & 3];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[1][1], 0, 0, 0);
        }
        if (c + 1 < nchunk) stash((c + 1) & 1);                            // (its readers of chunk c - 1 are behind this iteration's barrier)
    }
    // accumulator element i of lane (r, h): query row (i & 3) + 8 (i >> 2) + 4 h of the tile, candidate column r
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const long long col = x0 + wn + 32 * j2 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long row = q0 + wm + 32 * i2 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < rows_q && col < n) out[row * out_stride + col] = acc[i2][j2][i];
            }
        }
}

hipError_t dg_launch_sims_nt(const float* q, const float* x, long long rows_q, long long n, int F, long long q_stride, long long x_stride,
                             float* out, long long out_stride, hipStream_t s) {
    const dim3 grid((unsigned)((rows_q + SIMS_TM - 1) / SIMS_TM), (unsigned)((n + SIMS_TM - 1) / SIMS_TM));
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sims_nt, grid, dim3(256), 0, s, q, x, rows_q, n, F, q_stride, x_stride, out, out_stride);
    return hipGetLastError();
}

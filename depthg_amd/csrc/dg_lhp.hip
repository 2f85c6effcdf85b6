// Depth propagation of the LHP branch: LocalHiddenPositiveProjection.forward_depth before its
// projection head, reference src/modules.py:273-335:
//   pooled = adaptive_avg_pool2d(depth, (h,w)); points = depth2points(pooled, fov=90)        (:286-294)
//   dist[p][q] = |points_p - points_q|; row-wise min-max normalised; map = 1 - dist_n, zeroed where dist_n is above the
//   row's 1 % quantile (torch.quantile, linear interpolation)                                  (:297-319)
//   out[:, p] = mean_q map[p][q] * code[:, q]                                                  (:321-335)
// The (B,P,P) tensors of the reference are never formed: one wave per output position recomputes its P distances in
// registers, finds the quantile by removing the smallest value rank+1 times, and gathers the handful of code columns that
// survive.  The backward (adjoint: columns instead of rows) recomputes the same distances against the per-row statistics
// the forward stored - no neighbour lists, no atomics, fixed summation order.
//   k_lhp_points      (B,1,H,W) -> points (B,3,P), same arithmetic as the FPS sampler (explicit float32 operations)
//   k_lhp_propagate   forward (BWD = false) / backward (BWD = true); grid (ceil(P/4), B), block 256 = 4 waves
#include "dg_common.h"

// every float operation below is the reference's (numpy / torch CPU) operation, one rounding each: products that feed an
// addition go through dg_mul_rn (dg_common.h) - the __f*_rn intrinsics and the fp-contract pragma alone do not stop hipcc from
// fusing a*b+c

#define LHP_THREADS 256

__global__ __launch_bounds__(LHP_THREADS) void k_lhp_points(const float* __restrict__ depth, int H, int W, int h, int w, float factor,
                                                            float* __restrict__ points) {
    const int n = blockIdx.y, idx = blockIdx.x * LHP_THREADS + threadIdx.x, HW = h * w;
    if (idx >= HW) return;
    const float* d = depth + (size_t)n * H * W;
    const int i = idx / w, j = idx - i * w;
    const int ys = (i * H) / h, ye = ((i + 1) * H + h - 1) / h;
    const int xs = (j * W) / w, xe = ((j + 1) * W + w - 1) / w;
    float s = 0.f;                                       // row-major sequential sum, like the reference's CPU pooling
    for (int y = ys; y < ye; ++y)
        for (int x = xs; x < xe; ++x) s = __fadd_rn(s, d[(size_t)y * W + x]);
    const float dv = __fdiv_rn(__fdiv_rn(s, (float)(ye - ys)), (float)(xe - xs));     // sum / kh / kw: the operator's two divisions
    const float fd = __fmul_rn(factor, dv);
    float* p = points + (size_t)n * 3 * HW;
    p[idx] = __fdiv_rn(__fmul_rn(fd, __fsub_rn((float)j, (float)w / 2.0f)), (float)w);            // X
    p[HW + idx] = __fdiv_rn(__fmul_rn(fd, __fsub_rn((float)i, (float)h / 2.0f)), (float)h);       // Y
    p[2 * HW + idx] = __fmul_rn(-dv, 5.0f);                                                           // Z = -d * far
}

// correctly rounded float32 square root (through double: exact for sqrt), whatever the float32 sqrt lowering is
__device__ __forceinline__ float lhp_sqrt(float x) { return (float)sqrt((double)x); }

__device__ __forceinline__ float lhp_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float lhp_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// NJ = ceil(P / 64) rounded up to the template: every lane owns the partners lane + 64 j
template <int NJ, bool BWD>
__global__ __launch_bounds__(LHP_THREADS) void k_lhp_propagate(const float* __restrict__ src, const float* __restrict__ points,
                                                               float* __restrict__ stats, int D, int P, float* __restrict__ dst) {
    extern __shared__ float pts[];                        // [3][P]
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* pg = points + (size_t)n * 3 * P;
    for (int i = tid; i < 3 * P; i += LHP_THREADS) pts[i] = pg[i];
    __syncthreads();
    const int me = blockIdx.x * (LHP_THREADS / 64) + wid;               // forward: output row p; backward: code column q
    if (me >= P) return;
    const float mx_ = pts[me], my_ = pts[P + me], mz_ = pts[2 * P + me];
    const float infty = __builtin_inff();
    float wgt[NJ];                                        // weight of partner lane + 64 j, or -1 when it does not contribute
    float* st = stats + (size_t)n * P * 3;
    if (!BWD) {
        float dn[NJ];
        float mn = infty, mx = -infty;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            dn[j] = infty;
            if (q < P) {
                const float dx = __fsub_rn(mx_, pts[q]), dy = __fsub_rn(my_, pts[P + q]), dz = __fsub_rn(mz_, pts[2 * P + q]);
                dn[j] = lhp_sqrt(__fadd_rn(__fadd_rn(dg_mul_rn(dx, dx), dg_mul_rn(dy, dy)), dg_mul_rn(dz, dz)));
                mn = fminf(mn, dn[j]); mx = fmaxf(mx, dn[j]);
            }
        }
        mn = lhp_wave_min(mn); mx = lhp_wave_max(mx);
        const float range = __fsub_rn(mx, mn);
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (lane + 64 * j < P) dn[j] = __fdiv_rn(__fsub_rn(dn[j], mn), range);
        // the 1 % quantile: rank = 0.01 * (P - 1); the values at floor / ceil of it = remove the smallest ceil + 1 times
        const float rank = __fmul_rn(0.01f, (float)(P - 1));
        const int lo = (int)floorf(rank), hi = (int)ceilf(rank);
        unsigned long long gone = 0ull;                                   // bit j: partner j already removed
        float vlo = 0.f, vhi = 0.f;
        for (int rnd = 0; rnd <= hi; ++rnd) {
            float best = infty;
            int bj = -1;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (!((gone >> j) & 1ull) && dn[j] < best) { best = dn[j]; bj = j; }
            const float wmin = lhp_wave_min(best);
            const unsigned long long owners = __ballot(best == wmin && bj >= 0);
            if (owners == 0ull) break;                                    // (NaN rows: nothing comparable is left)
            if (lane == __ffsll((long long)owners) - 1) gone |= 1ull << bj;
            if (rnd == lo) vlo = wmin;
            if (rnd == hi) vhi = wmin;
        }
        const float w = __fsub_rn(rank, (float)lo);
        const float thr = w < 0.5f ? __fadd_rn(vlo, dg_mul_rn(w, __fsub_rn(vhi, vlo)))
                                   : __fsub_rn(vhi, dg_mul_rn(__fsub_rn(vhi, vlo), __fsub_rn(1.0f, w)));      // torch.lerp
        if (lane == 0) { st[me * 3] = mn; st[me * 3 + 1] = mx; st[me * 3 + 2] = thr; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) wgt[j] = (lane + 64 * j < P && !(dn[j] > thr)) ? __fsub_rn(1.0f, dn[j]) : -1.f;
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = lane + 64 * j;
            wgt[j] = -1.f;
            if (p < P) {
                const float dx = __fsub_rn(pts[p], mx_), dy = __fsub_rn(pts[P + p], my_), dz = __fsub_rn(pts[2 * P + p], mz_);
                const float dist = lhp_sqrt(__fadd_rn(__fadd_rn(dg_mul_rn(dx, dx), dg_mul_rn(dy, dy)), dg_mul_rn(dz, dz)));
                const float mn = st[p * 3], mx = st[p * 3 + 1], thr = st[p * 3 + 2];
                const float dnv = __fdiv_rn(__fsub_rn(dist, mn), __fsub_rn(mx, mn));
                if (!(dnv > thr)) wgt[j] = __fsub_rn(1.0f, dnv);
            }
        }
    }
    // gather: lanes run over the channels, the contributing partners are visited in ascending order
    float acc0 = 0.f, acc1 = 0.f;
    const float* sb = src + (size_t)n * D * P;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        unsigned long long m = __ballot(wgt[j] >= 0.f);
        while (m) {
            const int sl = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const float wq = __shfl(wgt[j], sl, 64);
            const int other = sl + 64 * j;
            if (lane < D) acc0 = __fadd_rn(acc0, __fmul_rn(wq, sb[(size_t)lane * P + other]));
            if (lane + 64 < D) acc1 = __fadd_rn(acc1, __fmul_rn(wq, sb[(size_t)(lane + 64) * P + other]));
        }
    }
    float* db = dst + (size_t)n * D * P;
    if (lane < D) db[(size_t)lane * P + me] = __fdiv_rn(acc0, (float)P);
    if (lane + 64 < D) db[(size_t)(lane + 64) * P + me] = __fdiv_rn(acc1, (float)P);
}

hipError_t dg_launch_lhp_points(const float* depth, int B, int H, int W, int h, int w, float factor, float* points, hipStream_t s) {
    hipLaunchKernelGGL(k_lhp_points, dim3((h * w + LHP_THREADS - 1) / LHP_THREADS, B), dim3(LHP_THREADS), 0, s, depth, H, W, h, w,
                       factor, points);
    return hipGetLastError();
}

hipError_t dg_launch_lhp_propagate(bool backward, const float* src, const float* points, float* stats, int B, int D, int P,
                                   float* dst, hipStream_t s) {
    const dim3 grid((P + 3) / 4, B), block(LHP_THREADS);
    const size_t smem = (size_t)3 * P * sizeof(float);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, block, smem, s, src, points, stats, D, P, dst);
        return hipGetLastError();
    };
    if (P <= 1024) return backward ? launch(k_lhp_propagate<16, true>) : launch(k_lhp_propagate<16, false>);
    if (P <= 4096) return backward ? launch(k_lhp_propagate<64, true>) : launch(k_lhp_propagate<64, false>);
    return hipErrorInvalidValue;
}

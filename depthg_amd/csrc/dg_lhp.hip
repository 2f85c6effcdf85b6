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

// ---- the other propagation maps of the LHP branch --------------------------------------------------------------------------
//   DG_LHP_ATTN        LocalHiddenPositiveProjection.forward_attn (src/modules.py:235-271): heads-mean of the last
//                      self-attention without the CLS row / column, row-wise min-max normalised, zero above the row's 99 %
//                      quantile, out[:, p] = mean_q map[p][q] code[:, q]
//   DG_LHP_ORIG_DEPTH  OriginalLocalHiddenPositiveProjection.forward_depth (:436-487): 1 - normalised point distance, zero
//                      where the distance is above the row mean, times the clipped 3x3 neighbourhood mask (:356-383),
//                      out[:, p] = sum_q map[p][q] code[:, q] / divide_num[p]
//   DG_LHP_ORIG_ATTN   OriginalLocalHiddenPositiveProjection.forward_attn (:403-434): heads-mean attention normalised by the
//                      row's 10 % / 90 % quantiles, zero below the row mean, same mask and divisor
// One wave per output row p, lanes over the partners q (register j of lane l is partner l + 64 j).  Order statistics come
// from a bitwise search on the monotone integer image of the floats (32 steps, counts through ballots: scalar work), so any
// rank costs the same.  ATTN keeps the (B,P,P) map for its backward when asked to (the attention tensor it is derived from is
// heads x larger); the Original variants keep their nine neighbour weights per row.
enum { LHP_ATTN = 0, LHP_ORIG_DEPTH = 1, LHP_ORIG_ATTN = 2 };

__device__ __forceinline__ float lhp_wave_sum(float v) {
#define LHP_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
    LHP_DPP_ADD(0x111, 0xf); LHP_DPP_ADD(0x112, 0xf); LHP_DPP_ADD(0x114, 0xf); LHP_DPP_ADD(0x118, 0xf);
    LHP_DPP_ADD(0x142, 0xa);
#undef LHP_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ uint32_t lhp_key(float x) {
    const uint32_t b = __float_as_uint(x);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float lhp_unkey(uint32_t k) { return __uint_as_float((k >> 31) ? (k ^ 0x80000000u) : ~k); }

// torch.quantile(row, q) with linear interpolation: the values at floor / ceil of q (P - 1) in sorted order, torch.lerp between
template <int NJ>
__device__ float lhp_quantile(const uint32_t (&key)[NJ], const int P, const float q) {
    const float rank = __fmul_rn(q, (float)(P - 1));
    const int lo = (int)floorf(rank), hi = (int)ceilf(rank);
    uint32_t res = 0u;
    for (int bit = 31; bit >= 0; --bit) {                 // largest x with #(key < x) <= lo: the lo-th smallest key
        const uint32_t cand = res | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) cnt += __popcll(__ballot(key[j] < cand));
        if (cnt <= lo) res = cand;
    }
    const float vlo = lhp_unkey(res);
    float vhi = vlo;
    if (hi != lo) {
        int le = 0;
        uint32_t nxt = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            le += __popcll(__ballot(key[j] <= res));
            if (key[j] > res && key[j] < nxt) nxt = key[j];
        }
        if (le < hi + 1) {                                // no duplicate of the lo-th value reaches rank hi: next larger key
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nxt, o, 64); nxt = t < nxt ? t : nxt; }
            vhi = lhp_unkey(nxt);
        }
    }
    const float w = __fsub_rn(rank, (float)lo);
    return w < 0.5f ? __fadd_rn(vlo, dg_mul_rn(w, __fsub_rn(vhi, vlo)))
                    : __fsub_rn(vhi, dg_mul_rn(__fsub_rn(vhi, vlo), __fsub_rn(1.0f, w)));                     // torch.lerp
}

struct LhpMapArgs {
    const float* code;       // (B,D,P)
    const float* attn;       // (B,heads,P+1,P+1)   ATTN / ORIG_ATTN
    const float* points;     // (B,3,P)             ORIG_DEPTH
    const float* divide;     // (P) divisors        ORIG_*
    float* out;              // (B,D,P)
    float* map;              // ATTN: (B,P,P); ORIG_*: (B,P,9)
    int D, P, heads, w;
};

template <int NJ, int MODE>
__global__ __launch_bounds__(LHP_THREADS) void k_lhp_map(const LhpMapArgs a) {
    extern __shared__ float lds[];                        // ORIG_DEPTH: points [3][P]; ORIG_*: then the waves' weight rows [4][P]
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, P = a.P, D = a.D;
    if (MODE == LHP_ORIG_DEPTH) {
        const float* pg = a.points + (size_t)n * 3 * P;
        for (int i = tid; i < 3 * P; i += LHP_THREADS) lds[i] = pg[i];
        __syncthreads();
    }
    const int me = blockIdx.x * (LHP_THREADS / 64) + wid;
    if (me >= P) return;
    const float infty = __builtin_inff();
    float v[NJ];
    // ---- the raw row
    if (MODE == LHP_ORIG_DEPTH) {
        const float mx_ = lds[me], my_ = lds[P + me], mz_ = lds[2 * P + me];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            v[j] = 0.f;
            if (q < P) {
                const float dx = __fsub_rn(mx_, lds[q]), dy = __fsub_rn(my_, lds[P + q]), dz = __fsub_rn(mz_, lds[2 * P + q]);
                v[j] = lhp_sqrt(__fadd_rn(__fadd_rn(dg_mul_rn(dx, dx), dg_mul_rn(dy, dy)), dg_mul_rn(dz, dz)));
            }
        }
    } else {
        const size_t P1 = (size_t)P + 1;
        const float* ar = a.attn + ((size_t)n * a.heads * P1 + (size_t)(me + 1)) * P1 + 1;      // head 0, row me + 1, column 1
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = 0.f;
        const size_t hs = P1 * P1;
        int hd = 0;
        for (; hd + 3 <= a.heads; hd += 3) {              // torch.mean(dim=1): heads summed in order, one division; three heads'
            float t0[NJ], t1[NJ], t2[NJ];                 // loads in flight
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                t0[j] = t1[j] = t2[j] = 0.f;
                if (q < P) {
                    t0[j] = __builtin_nontemporal_load(ar + (size_t)hd * hs + q);
                    t1[j] = __builtin_nontemporal_load(ar + (size_t)(hd + 1) * hs + q);
                    t2[j] = __builtin_nontemporal_load(ar + (size_t)(hd + 2) * hs + q);
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[j] = __fadd_rn(__fadd_rn(__fadd_rn(v[j], t0[j]), t1[j]), t2[j]);
        }
        for (; hd < a.heads; ++hd) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                if (q < P) v[j] = __fadd_rn(v[j], __builtin_nontemporal_load(ar + (size_t)hd * hs + q));
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = __fdiv_rn(v[j], (float)a.heads);
    }
    // ---- normalisation
    float lo, hi;
    if (MODE == LHP_ORIG_ATTN) {
        uint32_t key[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) key[j] = lane + 64 * j < P ? lhp_key(v[j]) : 0xffffffffu;
        hi = lhp_quantile<NJ>(key, P, 0.9f);
        lo = lhp_quantile<NJ>(key, P, 0.1f);
    } else {
        lo = infty; hi = -infty;
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (lane + 64 * j < P) { lo = fminf(lo, v[j]); hi = fmaxf(hi, v[j]); }
        lo = lhp_wave_min(lo); hi = lhp_wave_max(hi);
    }
    const float range = __fsub_rn(hi, lo);
#pragma unroll
    for (int j = 0; j < NJ; ++j) if (lane + 64 * j < P) v[j] = __fdiv_rn(__fsub_rn(v[j], lo), range);
    // ---- threshold -> weights
    if (MODE == LHP_ATTN) {
        uint32_t key[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) key[j] = lane + 64 * j < P ? lhp_key(v[j]) : 0xffffffffu;
        const float thr = lhp_quantile<NJ>(key, P, 0.99f);
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = (lane + 64 * j < P && !(v[j] > thr)) ? v[j] : 0.f;
    } else {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (lane + 64 * j < P) sum += v[j];
        const float mean = lhp_wave_sum(sum) / (float)P;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (MODE == LHP_ORIG_DEPTH) v[j] = (lane + 64 * j < P && !(v[j] > mean)) ? __fsub_rn(1.0f, v[j]) : 0.f;
            else v[j] = (lane + 64 * j < P && !(v[j] < mean)) ? v[j] : 0.f;
        }
    }
    const float* cb = a.code + (size_t)n * D * P;
    float* ob = a.out + (size_t)n * D * P;
    if (MODE == LHP_ATTN) {                               // the weighted mean is k_lhp_attn_apply's
        float* mr = a.map + ((size_t)n * P + me) * P;
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (lane + 64 * j < P) mr[lane + 64 * j] = v[j];
    } else {
        // the mask leaves the clipped 3x3 neighbourhood: nine weights per row through LDS, lanes over the channels
        float* wrow = lds + (MODE == LHP_ORIG_DEPTH ? 3 * P : 0) + wid * P;
#pragma unroll
        for (int j = 0; j < NJ; ++j) if (lane + 64 * j < P) wrow[lane + 64 * j] = v[j];
        __builtin_amdgcn_wave_barrier();
        const int W = a.w, H = P / W, pi = me / W, pj = me - pi * W;
        float w9 = 0.f;
        int q9 = -1;
        if (lane < 9) {
            const int qi = pi + lane / 3 - 1, qj = pj + lane % 3 - 1;
            if (qi >= 0 && qi < H && qj >= 0 && qj < W) { q9 = qi * W + qj; w9 = wrow[q9]; }
            a.map[((size_t)n * P + me) * 9 + lane] = w9;
        }
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int q = __shfl(q9, t, 64);
            const float wq = __shfl(w9, t, 64);
            if (q < 0) continue;
            if (lane < D) acc0 += wq * cb[(size_t)lane * P + q];
            if (lane + 64 < D) acc1 += wq * cb[(size_t)(lane + 64) * P + q];
        }
        const float dv = a.divide[me];
        if (lane < D) ob[(size_t)lane * P + me] = __fdiv_rn(acc0, dv);
        if (lane + 64 < D) ob[(size_t)(lane + 64) * P + me] = __fdiv_rn(acc1, dv);
    }
}

// DG_LHP_ATTN, the weighted means over the (B,P,P) map:  forward (TR)  out[d][p]       = (1/P) sum_q map[p][q] code[d][q]
//                                                      backward      grad_code[d][q] = (1/P) sum_p map[p][q] g[d][p]
// Block = 64 output positions x (lanes) x all contraction positions c in chunks of 64: per chunk the block stages the source
// rows transposed ([c][D], read as broadcast float4) and the 64 x 64 map tile (coalesced rows; the forward reads it
// transposed, row stride 65) in LDS; wave wv takes the chunk's positions = wv (mod 4); the four partial accumulators meet in
// LDS in a fixed order.  ND = ceil(D / 4) float4 groups.
template <int ND, bool TR>
__global__ __launch_bounds__(LHP_THREADS) void k_lhp_attn_apply(const float* __restrict__ src, const float* __restrict__ map, int D, int P,
                                                                float* __restrict__ dst) {
    extern __shared__ float lds[];                        // [64][4 ND + 4] source rows, [64][65] map tile; later [4][4 ND][64] partial sums
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, x0 = blockIdx.x * 64;
    constexpr int DP = 4 * ND, RS = DP + 4;               // (row stride: staging stores spread over 16 banks)
    float* tile = lds + 64 * RS;
    float acc[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) acc[d] = 0.f;
    const float* sb = src + (size_t)n * D * P;
    const float* mb = map + (size_t)n * P * P;
    for (int c0 = 0; c0 < P; c0 += 64) {
        float tv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {                    // tile row = wid * 16 + i: map row (TR ? x0 : c0) + row, 64 columns from (TR ? c0 : x0)
            const int gr = (TR ? x0 : c0) + wid * 16 + i, gc = (TR ? c0 : x0) + lane;
            tv[i] = (gr < P && gc < P) ? mb[(size_t)gr * P + gc] : 0.f;
        }
        float sv[ND];                                     // source element (d = 4 i + wid, position c0 + lane): coalesced along the positions
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d = 4 * i + wid;
            sv[i] = (d < D && c0 + lane < P) ? sb[(size_t)d * P + c0 + lane] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ND; ++i) lds[lane * RS + 4 * i + wid] = sv[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) tile[(wid * 16 + i) * 65 + lane] = tv[i];
        __syncthreads();
        const int rows = P - c0 < 64 ? P - c0 : 64;
        for (int r = wid; r < rows; r += 4) {
            const float m = TR ? tile[lane * 65 + r] : tile[r * 65 + lane];
            const float4* gr = reinterpret_cast<const float4*>(lds + r * RS);
            float4 gv[ND];
#pragma unroll
            for (int k = 0; k < ND; ++k) gv[k] = gr[k];    // all broadcast reads of the row in flight before the first multiply
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                acc[4 * k] = fmaf(m, gv[k].x, acc[4 * k]); acc[4 * k + 1] = fmaf(m, gv[k].y, acc[4 * k + 1]);
                acc[4 * k + 2] = fmaf(m, gv[k].z, acc[4 * k + 2]); acc[4 * k + 3] = fmaf(m, gv[k].w, acc[4 * k + 3]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DP; ++d) lds[(wid * DP + d) * 64 + lane] = acc[d];
    __syncthreads();
    for (int i = tid; i < DP * 64; i += LHP_THREADS) {
        const int d = i >> 6, l = i & 63, x = x0 + l;
        if (d < D && x < P) {
            const float s = ((lds[(0 * DP + d) * 64 + l] + lds[(1 * DP + d) * 64 + l]) + lds[(2 * DP + d) * 64 + l]) + lds[(3 * DP + d) * 64 + l];
            dst[((size_t)n * D + d) * P + x] = s / (float)P;
        }
    }
}

// backward of the Original variants: grad_code[d][q] = sum over the rows p of q's clipped 3x3 neighbourhood of
// map9[p][slot of q in p's neighbourhood] * (g[d][p] / divide_num[p]); one thread per (d, q)
__global__ __launch_bounds__(LHP_THREADS) void k_lhp_local_bwd(const float* __restrict__ g, const float* __restrict__ map9,
                                                               const float* __restrict__ divide, int D, int P, int W,
                                                               float* __restrict__ gcode) {
    const int n = blockIdx.z, d = blockIdx.y, q = blockIdx.x * LHP_THREADS + threadIdx.x;
    if (q >= P) return;
    const int H = P / W, qi = q / W, qj = q - qi * W;
    const float* gr = g + ((size_t)n * D + d) * P;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {                         // row p sees q at slot t  <=>  q = p + (t/3 - 1, t%3 - 1)
        const int pi = qi - (t / 3 - 1), pj = qj - (t % 3 - 1);
        if (pi < 0 || pi >= H || pj < 0 || pj >= W) continue;
        const int p = pi * W + pj;
        acc += map9[((size_t)n * P + p) * 9 + t] * __fdiv_rn(gr[p], divide[p]);
    }
    gcode[((size_t)n * D + d) * P + q] = acc;
}

static hipError_t lhp_attn_apply(bool forward, const float* src, const float* map, int B, int D, int P, float* dst, hipStream_t s);

hipError_t dg_launch_lhp_map(int mode, const float* code, const float* attn, const float* points, const float* divide, int B, int D,
                             int h, int w, int heads, float* out, float* map, hipStream_t s) {
    const int P = h * w;
    LhpMapArgs a{code, attn, points, divide, out, map, D, P, heads, w};
    const dim3 grid((P + 3) / 4, B), block(LHP_THREADS);
    const size_t smem = mode == LHP_ATTN ? 0 : (size_t)(mode == LHP_ORIG_DEPTH ? 7 : 4) * P * sizeof(float);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, block, smem, s, a);
        return hipGetLastError();
    };
    if (mode == LHP_ATTN) {
        const hipError_t e = P <= 1024 ? launch(k_lhp_map<16, LHP_ATTN>) : P <= 4096 ? launch(k_lhp_map<64, LHP_ATTN>) : hipErrorInvalidValue;
        return e != hipSuccess ? e : lhp_attn_apply(true, code, map, B, D, P, out, s);
    }
    if (P <= 1024) {
        if (mode == LHP_ORIG_DEPTH) return launch(k_lhp_map<16, LHP_ORIG_DEPTH>);
        if (mode == LHP_ORIG_ATTN) return launch(k_lhp_map<16, LHP_ORIG_ATTN>);
    } else if (P <= 4096) {
        if (mode == LHP_ORIG_DEPTH) return launch(k_lhp_map<64, LHP_ORIG_DEPTH>);
        if (mode == LHP_ORIG_ATTN) return launch(k_lhp_map<64, LHP_ORIG_ATTN>);
    }
    return hipErrorInvalidValue;
}

static hipError_t lhp_attn_apply(bool forward, const float* src, const float* map, int B, int D, int P, float* dst, hipStream_t s) {
    const dim3 grid((P + 63) / 64, B), block(LHP_THREADS);
    auto launch = [&](auto kern, int nd) -> hipError_t {
        const int dp = 4 * nd, work = 64 * (dp + 4) + 64 * 65, fin = 4 * dp * 64;
        const size_t smem = (size_t)(work > fin ? work : fin) * sizeof(float);
        hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, block, smem, s, src, map, D, P, dst);
        return hipGetLastError();
    };
    if (forward) {
        if (D <= 32) return launch(k_lhp_attn_apply<8, true>, 8);
        if (D <= 64) return launch(k_lhp_attn_apply<16, true>, 16);
        if (D <= 96) return launch(k_lhp_attn_apply<24, true>, 24);
        if (D <= 128) return launch(k_lhp_attn_apply<32, true>, 32);
    } else {
        if (D <= 32) return launch(k_lhp_attn_apply<8, false>, 8);
        if (D <= 64) return launch(k_lhp_attn_apply<16, false>, 16);
        if (D <= 96) return launch(k_lhp_attn_apply<24, false>, 24);
        if (D <= 128) return launch(k_lhp_attn_apply<32, false>, 32);
    }
    return hipErrorInvalidValue;
}

hipError_t dg_launch_lhp_map_bwd(int mode, const float* g, const float* map, const float* divide, int B, int D, int h, int w,
                                 float* gcode, hipStream_t s) {
    const int P = h * w;
    if (mode == LHP_ATTN) return lhp_attn_apply(false, g, map, B, D, P, gcode, s);
    hipLaunchKernelGGL(k_lhp_local_bwd, dim3((P + LHP_THREADS - 1) / LHP_THREADS, D, B), dim3(LHP_THREADS), 0, s, g, map, divide, D, P, w,
                       gcode);
    return hipGetLastError();
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

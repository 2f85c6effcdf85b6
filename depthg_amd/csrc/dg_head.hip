// The segmentation head of DinoFeaturizer (src/modules.py:75-88, 122-137) as gfx950 kernels:
//     code = cluster1(drop1(f)) + cluster2(drop2(f)),   cluster1 = conv1x1(C -> D),  cluster2 = conv1x1(C -> C), ReLU, conv1x1(C -> D)
//     feats = drop3(f)                                    (Dropout2d(p): whole channels of an image zeroed, the rest scaled by 1/(1-p))
// One fused forward launch per call: a block owns NT positions of one image, reads its fp32 feature tile ONCE (writing drop3(f) on
// the way), keeps it as a bf16 [channel][position] image in LDS, runs the hidden 1x1 convolution on the matrix cores (bf16 in,
// fp32 accumulate; the channel-major tile is the k-strided operand, read with the transposing ds_read_b64_tr_b16), applies bias +
// ReLU in the accumulators, keeps the hidden tile in LDS as the next product's operand and runs both output convolutions.  Dropout2d
// costs nothing: a dropped channel is a zeroed COLUMN of the weights (exact), the 1/(1-p) factor is applied to the fp32 accumulator.
// Backward (the ViT is frozen: gradients for the six head tensors only): k_head_dh (d hidden from d code, ReLU mask), k_head_wgrad
// (the three weight gradients: products over all positions of the batch, split over blocks, partial sums reduced in a fixed order),
// k_head_rowsum (bias gradients).  No floating-point atomics: results are bit-reproducible.
#include "dg_common.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;


__device__ __forceinline__ bf16x8 pack8(const f32x4 lo, const f32x4 hi) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)lo[e]; o[4 + e] = (__bf16)hi[e]; }
    return o;
}

// B (or A) fragment of a 16x16x32 MFMA from a [k][n] bf16 image in LDS (row stride `rowb` bytes): the lanes of group g = lane>>4
// hold k = k0 + 8g .. 8g+7 of column n0 + (lane & 15).  Two transposing reads of 4 rows each; odd groups take their two row
// blocks in the opposite order so that the 32 lanes of a half hit 8 rows whose 32-byte pieces fall into different banks (row
// stride = 5 or 3 times 32 bytes) - the other operand of the MFMA uses the same k order (frag_k_order).
__device__ __forceinline__ bf16x8 tr_frag(const char* img, const int rowb, const int k0, const int n0, const int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3, sw = g & 1;
    bf16x8 o;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = k0 + 8 * g + 4 * (u ^ sw) + q;
        const s16x4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + row * rowb + (n0 + 4 * pc) * 2));
        const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * u + e] = tb[e];
    }
    return o;
}
// first k of element block u (0: elements 0..3, 1: elements 4..7) of a lane's fragment in the order tr_frag delivers
__device__ __forceinline__ int frag_k_order(const int lane, const int u) { return 8 * (lane >> 4) + 4 * (u ^ ((lane >> 4) & 1)); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Raw A fragment from bf16 weights W[row][kbase + 8g .. + 7]: one 16-byte load from a clamped (always valid) address.  K is a
// multiple of 8.  NO select on the loaded value: a VALU operation on it right behind the load makes hipcc wait for the load where it
// is issued - the k-step-ahead prefetch of the GEMM loops then overlaps nothing (2.0 k cycles per k-step against 0.9 k of MFMAs).
// Rows beyond the matrix and k beyond K therefore deliver (finite) values of other rows / columns: k >= K meets zero keep bits or a
// zero operand, rows >= nrows produce accumulator rows that are never stored.
__device__ __forceinline__ u32x4 wraw(const __bf16* __restrict__ W, const int ld, const int row, const int nrows, const int kbase,
                                      const int K, const int lane) {
    const int k = kbase + 8 * (lane >> 4);
    return *reinterpret_cast<const u32x4*>(W + (size_t)(row < nrows ? row : nrows - 1) * ld + (k < K ? k : K - 8));
}
// ... masked with the keep bits of its eight k (16-bit all-ones / zero words, natural k order) and brought into tr_frag's k order
// (odd lane groups hold their two blocks of four in the opposite order)
__device__ __forceinline__ bf16x8 wfrag(u32x4 v, const u32x4 keep, const int lane) {
    v &= keep;
    if ((lane >> 4) & 1) v = u32x4{v[2], v[3], v[0], v[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// fp32 parameters -> bf16 copies for the MFMA kernels (DgHeadWeightLayout, dg_common.h): w1 (D,C), w2a (C,C), w2b (D,C) fragment-major
// for the forward, and w2b transposed (C, DP) row-major with DP = D rounded up to 32 (zero padded) for the backward's d hidden product
__global__ __launch_bounds__(256) void k_head_prep(const float* __restrict__ w1, const float* __restrict__ w2a, const float* __restrict__ w2b,
                                                   __bf16* __restrict__ scratch, int C, int D, int DP) {
    const DgHeadWeightLayout L(C, D);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // tile-major index -> (row, k): [block rb][k-step ks][lane][e], lane = 16 g + r: row = 16 rb + r, k = 32 ks + 8 g + e
    auto rk = [&](size_t t, int& row, int& k) {
        const int e = (int)(t & 7), lane = (int)((t >> 3) & 63);
        const size_t tile = t >> 9;
        const int ks = (int)(tile % L.KS), rb = (int)(tile / L.KS);
        row = 16 * rb + (lane & 15); k = 32 * ks + 8 * (lane >> 4) + e;
    };
    if (i < L.w2a) {                                        // w1: eight row blocks
        int row, k; rk(i, row, k);
        scratch[L.w1 + i] = (row < D && k < C) ? (__bf16)w1[(size_t)row * C + k] : (__bf16)0.f;
        if (w2b) scratch[L.w2b + i] = (row < D && k < C) ? (__bf16)w2b[(size_t)row * C + k] : (__bf16)0.f;
    }
    if (w2a && i < L.w2b - L.w2a) {
        int row, k; rk(i, row, k);
        scratch[L.w2a + i] = (row < C && k < C) ? (__bf16)w2a[(size_t)row * C + k] : (__bf16)0.f;
    }
    if (w2b && i < (size_t)C * DP) { const int m = (int)(i / DP), d = (int)(i - (size_t)m * DP); scratch[L.w2bT + i] = d < D ? (__bf16)w2b[(size_t)d * C + m] : (__bf16)0.f; }
}
hipError_t dg_launch_head_prep(const float* w1, const float* w2a, const float* w2b, void* scratch, int C, int D, hipStream_t s) {
    const int DP = (D + 31) / 32 * 32;
    const DgHeadWeightLayout L(C, D);
    size_t n = L.w2a;                                       // (w1 / w2b tiles)
    if (w2a && L.w2b - L.w2a > n) n = L.w2b - L.w2a;
    if (w2b && (size_t)C * DP > n) n = (size_t)C * DP;
    hipLaunchKernelGGL(k_head_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w1, w2a, w2b, static_cast<__bf16*>(scratch), C, D, DP);
    return hipGetLastError();
}
// A fragment of row block rb, k-step ks from a fragment-major copy: one 16-byte load per lane, 1 KiB contiguous per wave.  NO select on
// the loaded value (see wraw above); the padding is zeros in memory
__device__ __forceinline__ u32x4 wtile(const __bf16* __restrict__ Wt, const int KS, const int rb, const int ks, const int lane) {
    return *reinterpret_cast<const u32x4*>(Wt + (((size_t)rb * KS + ks) * 64 + lane) * 8);
}

// MB = 16-row blocks of hidden channels per wave, NW = waves per block (Cpad = 16 MB NW), NT = positions per block
// (head_frow(NT): an odd number of 32-byte pieces per row, what tr_frag's bank pattern needs)
__host__ __device__ constexpr int head_frow(int NT) { return ((NT * 2 / 32) & 1) ? NT * 2 : NT * 2 + 32; }
#ifndef HEAD_FWD_PD
#define HEAD_FWD_PD 3
#endif
template <int MB, int NT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_head_fwd(const DgHeadFwdArgs a) {
    constexpr int NTH = 64 * NW, CP = 16 * MB * NW, NB = NT / 16, FROW = head_frow(NT), DBMAX = 8, KS = CP / 32;
    static_assert(NT % 16 == 0 && DBMAX % NW == 0 && CP % 32 == 0, "blocking");
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    char* const Ft = hsm;                                   // [CP][FROW]  bf16 f tile, channel-major; later the hidden tile, same layout
    unsigned short* const km1 = reinterpret_cast<unsigned short*>(Ft + CP * FROW);   // [CP] keep bits of cluster1's dropout (0xffff / 0)
    unsigned short* const km2 = km1 + CP;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.y, p0 = blockIdx.x * NT;
    const int C = a.C, D = a.D, P = a.P;
    const bool nonlinear = a.w2a != nullptr;
    const float s1 = a.keep1 ? a.scale : 1.f, s2 = a.keep2 ? a.scale : 1.f, s3 = a.keep3 ? a.scale : 1.f;
#ifdef DG_DEVTOOLS
#define HSTAMP(k) if (a.stamps && blockIdx.x == 3 && blockIdx.y == 5 && tid == 0) a.stamps[k] = __builtin_amdgcn_s_memtime();
#else
#define HSTAMP(k)
#endif
    HSTAMP(0)

    // ---- stage the feature tile: fp32 (C, NT) -> bf16 LDS image, drop3(f) written on the way.  Eight rows' loads in flight per
    //      thread (one block per CU: nothing else hides the latency)
    // keep flags of the two dropouts: requested here, unconditionally (a null mask reads any valid word), and turned into the bit masks
    // BEHIND the tile's loads - under `!keep || keep[k] != 0` each was a branch with a load and its own wait: two memory round trips
    // (2-4 us under load) before the tile's first load was issued
    constexpr int KIT = (CP + NTH - 1) / NTH;
    float rk1[KIT], rk2[KIT];
    {
        const float* const p1 = a.keep1 ? a.keep1 + (size_t)b * C : a.feat;
        const float* const p2 = a.keep2 ? a.keep2 + (size_t)b * C : a.feat;
#pragma unroll
        for (int u = 0; u < KIT; ++u) {
            const int k = tid + NTH * u, kc = k < C ? k : C - 1;
            rk1[u] = p1[kc];
            rk2[u] = p2[kc];
        }
    }
    {
        constexpr int Q = NT / 4, NIT = (CP * Q + NTH - 1) / NTH;        // 16-byte pieces per row; pieces per thread
        // loads in flight per thread: all of a thread's pieces where they fit the registers (the accumulators are not live yet)
        constexpr int U = NIT <= 24 ? NIT : (NIT % 8 == 0 ? 8 : (NIT % 6 == 0 ? 6 : (NIT % 4 == 0 ? 4 : (NIT % 3 == 0 ? 3 : 1))));
        if ((P & 3) == 0) {
            // fast path: every load unconditional from a clamped address (a load under a condition becomes a branch with its own
            // wait), values selected afterwards; U pieces in flight per thread
            const float* const k3p = a.keep3 ? a.keep3 : a.feat;       // (no dropout of the features: any valid address, value unused)
            for (int it = 0; it < NIT; it += U) {
                f32x4 v[U];
                float kf3[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = (it + u) * NTH + tid, k = i / Q, q4 = i - k * Q, p = p0 + 4 * q4;
                    const int kc = k < C ? k : C - 1, pc = p + 3 < P ? p : P - 4;
                    v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.feat + dg_img_off(b, (long long)C * P, a.Bs, a.d_feat) + (size_t)kc * P + pc));
                    kf3[u] = k3p[(size_t)b * C + kc];        // (raw: an operation on a loaded value here makes hipcc wait for ALL loads issued so far)
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = (it + u) * NTH + tid, k = i / Q, q4 = i - k * Q, p = p0 + 4 * q4;
                    const bool ok = k < C && p + 3 < P;
                    const f32x4 vv = ok ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
                    if (a.feats_out && ok) *reinterpret_cast<f32x4*>(a.feats_out + dg_img_off(b, (long long)C * P, a.Bs, a.d_fo) + (size_t)k * P + p) = vv * (a.keep3 ? kf3[u] * s3 : 1.f);
                    bf16x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (__bf16)vv[e];
                    if (k < CP) *reinterpret_cast<bf16x4*>(Ft + k * FROW + q4 * 8) = o4;
                }
            }
        } else {
            // position counts that are not a multiple of 4 (odd maps): element-wise, guarded
            for (int i = tid; i < CP * Q; i += NTH) {
                const int k = i / Q, q4 = i - k * Q, p = p0 + 4 * q4;
                bf16x4 o4;
                const float f3 = (a.keep3 && k < C) ? a.keep3[(size_t)b * C + k] * s3 : 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (k < C && p + e < P) ? a.feat[dg_img_off(b, (long long)C * P, a.Bs, a.d_feat) + (size_t)k * P + p + e] : 0.f;
                    if (a.feats_out && k < C && p + e < P) a.feats_out[dg_img_off(b, (long long)C * P, a.Bs, a.d_fo) + (size_t)k * P + p + e] = v * f3;
                    o4[e] = (__bf16)v;
                }
                *reinterpret_cast<bf16x4*>(Ft + k * FROW + q4 * 8) = o4;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < KIT; ++u) {
        const int k = tid + NTH * u;
        if (k < CP) {
            km1[k] = (k < C && (!a.keep1 || rk1[u] != 0.f)) ? 0xffffu : 0u;
            km2[k] = (k < C && (!a.keep2 || rk2[u] != 0.f)) ? 0xffffu : 0u;
        }
    }
    __syncthreads();
    HSTAMP(1)

    // ---- one k-loop over the input channels for both products that read the feature tile:
    //        hidden_pre = W2a[:, kept2] f   (wave `wid` owns hidden channels [16 MB wid, 16 MB (wid + 1)), all NB position blocks)
    //        code1      = W1[:, kept1] f    (wave `wid` owns code-channel blocks wid and wid + 4, all NB position blocks)
    //      The weight fragments of k-step ks + 1 are in flight while step ks runs.
    f32x4 acc1[MB][NB];
    const int mbase = wid * 16 * MB;
    // the two output convolutions: wave `wid` owns code-channel blocks wid + NW i (16 channels each, D <= 128) for ALL position
    // blocks - every weight row is fetched by exactly one wave of the block
    constexpr int NA = DBMAX / NW;
    f32x4 acc2a[NA][NB], acc2b[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc2a[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        // weight fragments PD k-steps ahead: one step of 28 MFMAs (0.2 us) does not cover an L2 round trip (~1 us) - three ahead where the
        // k-steps divide by three (the 384-channel forms: 96 KB of fragments in flight per CU instead of 64), two elsewhere
        constexpr int PD = HEAD_FWD_PD > 2 && KS % 3 == 0 ? 3 : 2;
        static_assert(KS % PD == 0, "PD k-steps per iteration");
        u32x4 wn[PD][MB], vn[PD][NA];
        auto fetch = [&](const int ks, const int q) {
            if (nonlinear) {
#pragma unroll
                for (int i = 0; i < MB; ++i) wn[q][i] = wtile(a.w2a_bf, KS, wid * MB + i, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) vn[q][i] = wtile(a.w1_bf, KS, wid + NW * i, ks, lane);
        };
#pragma unroll
        for (int q = 0; q < PD; ++q) fetch(q, q);
        for (int ks0 = 0; ks0 < KS; ks0 += PD) {
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                const int ks = ks0 + q;
                u32x4 wc[MB], vc[NA];
#pragma unroll
                for (int i = 0; i < MB; ++i) wc[i] = wn[q][i];
#pragma unroll
                for (int i = 0; i < NA; ++i) vc[i] = vn[q][i];
                if (ks + PD < KS) fetch(ks + PD, q);
                const u32x4 keep2 = *reinterpret_cast<const u32x4*>(km2 + 32 * ks + 8 * g);
                const u32x4 keep1 = *reinterpret_cast<const u32x4*>(km1 + 32 * ks + 8 * g);
                bf16x8 bfr[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) bfr[j] = tr_frag(Ft, FROW, 32 * ks, 16 * j, lane);
                if (nonlinear) {
#pragma unroll
                    for (int i = 0; i < MB; ++i) {
                        const bf16x8 af = wfrag(wc[i], keep2, lane);
#pragma unroll
                        for (int j = 0; j < NB; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc1[i][j], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < NA; ++i) {    // (unconditional: rows beyond D are zero fragments; a condition around an MFMA makes hipcc
                                                  //  shuffle the whole accumulator array through v_accvgpr moves at every branch)
                    const bf16x8 af = wfrag(vc[i], keep1, lane);
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc2a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc2a[i][j], 0, 0, 0);
                }
            }
        }
    }
    HSTAMP(2)
    // the output convolution's weight fragments, all k-steps, where they fit (the hidden accumulators die in the next phase): requested
    // here, they arrive under the bias / ReLU / hidden-image phases
    constexpr bool W2B_ALL = NA * KS <= 12;
    u32x4 w2all[W2B_ALL ? KS : 1][NA];
    if constexpr (W2B_ALL) {
        if (nonlinear) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < NA; ++i) w2all[ks][i] = wtile(a.w2b_bf, KS, wid + NW * i, ks, lane);
        }
    }
    __syncthreads();                                       // every wave is done reading the f tile
    HSTAMP(3)
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc2b[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};     // (first touched here: not live during the k-loop above)
    if (nonlinear) {
        // bias + ReLU in the accumulators; hidden tile -> the channel-major image the f tile occupied (the next product reads it
        // with the same transposing fragment reads; it is also what goes to HBM)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int m = mbase + 16 * i + 4 * g;
            f32x4 bias;
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = a.b2a[m + r < C ? m + r : C - 1];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int p = 16 * j + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = m + r < C ? fmaxf(fmaf(acc1[i][j][r], s2, bias[r]), 0.f) : 0.f;
                    *reinterpret_cast<__bf16*>(Ft + (m + r) * FROW + p * 2) = (__bf16)h;
                }
            }
        }
        __syncthreads();
        HSTAMP(4)
        // ---- cluster2's output convolution: W2b hidden
        const u32x4 ones = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if constexpr (W2B_ALL) {
            // (every k-step's fragment was requested behind the first k-loop, w2all: this loop is MFMAs and LDS reads only - one
            //  fragment per wave and k-step two steps ahead was a chain of L2 round trips, 7 k cycles for 84 MFMAs)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 bfr[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) bfr[j] = tr_frag(Ft, FROW, 32 * ks, 16 * j, lane);
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const bf16x8 af = wfrag(w2all[ks][i], ones, lane);
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc2b[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc2b[i][j], 0, 0, 0);
                }
            }
        } else {
        static_assert(KS % 2 == 0, "two k-steps per iteration");
        u32x4 wn[2][NA];
        auto fetch2 = [&](const int ks) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int i = 0; i < NA; ++i) wn[h2][i] = wtile(a.w2b_bf, KS, wid + NW * i, ks + h2, lane);
        };
        fetch2(0);
        for (int ks = 0; ks < KS; ks += 2) {
            u32x4 wc[2][NA];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int i = 0; i < NA; ++i) wc[h2][i] = wn[h2][i];
            if (ks + 2 < KS) fetch2(ks + 2);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                bf16x8 bfr[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) bfr[j] = tr_frag(Ft, FROW, 32 * (ks + h2), 16 * j, lane);
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const bf16x8 af = wfrag(wc[h2][i], ones, lane);
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc2b[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc2b[i][j], 0, 0, 0);
                }
            }
        }
        }
    }
    HSTAMP(5)
    // ---- code = s1 * (W1 f) + (W2b hidden) + b1 + b2b
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float bb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 16 * (wid + NW * i) + 4 * g + r, dc = d < D ? d : D - 1;
            bb[r] = a.b1[dc] + (nonlinear ? a.b2b[dc] : 0.f);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int p = p0 + 16 * j + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * (wid + NW * i) + 4 * g + r;
                if (d < D && p < P) a.code[dg_img_off(b, (long long)D * P, a.Bs, a.d_code) + (size_t)d * P + p] = fmaf(acc2a[i][j][r], s1, bb[r]) + (nonlinear ? acc2b[i][j][r] : 0.f);
            }
        }
    }
    HSTAMP(6)
    // ---- hidden tile -> HBM (B, C, P) bf16, rows copied from the channel-major LDS image
    if (nonlinear && a.hidden) {
        constexpr int Q8 = NT / 8;
        const bool vec = (P & 7) == 0;
        for (int idx = tid; idx < C * Q8; idx += NTH) {
            const int m = idx / Q8, c = idx - m * Q8, p = p0 + 8 * c;
            __bf16* dst = a.hidden + ((size_t)b * C + m) * P + p;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(Ft + m * FROW + c * 16);
            if (vec && p + 7 < P) *reinterpret_cast<bf16x8*>(dst) = v;
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (p + e < P) dst[e] = v[e];
            }
        }
    }
    HSTAMP(7)
}

template <int MB, int NT, int NW = 4>
static hipError_t launch_head_fwd(const DgHeadFwdArgs& a, hipStream_t s) {
    constexpr int CP = 16 * MB * NW;
    const int smem = CP * head_frow(NT) + 2 * CP * 2;
    auto kern = k_head_fwd<MB, NT, NW>;
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
    if (e != hipSuccess) return e;
#ifdef DG_DEVTOOLS
    if (const char* f = getenv("DG_HEAD_STAMPS")) {
        static unsigned long long* buf = nullptr;
        if (!buf && hipMalloc(&buf, 64) != hipSuccess) return hipErrorOutOfMemory;
        DgHeadFwdArgs a2 = a;
        a2.stamps = buf;
        hipLaunchKernelGGL(kern, dim3((a.P + NT - 1) / NT, a.B), dim3(64 * NW), smem, s, a2);
        unsigned long long h[8];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, buf, 64, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* fp = fopen(f, "w")) { for (int i = 1; i < 8; ++i) fprintf(fp, "phase %d: %llu cycles\n", i, h[i] - h[i - 1]); fclose(fp); }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(kern, dim3((a.P + NT - 1) / NT, a.B), dim3(64 * NW), smem, s, a);
    return hipGetLastError();
}

hipError_t dg_launch_head_fwd(const DgHeadFwdArgs& a, hipStream_t s) {
    if (a.C <= 64) return launch_head_fwd<1, 64>(a, s);
    if (a.C <= 128) return launch_head_fwd<2, 64>(a, s);
    if (a.C <= 192) return launch_head_fwd<3, 64>(a, s);
    // C <= 384 on maps whose positions split into tiles of 112 (28 x 28, 56 x 56, ...): eight waves and 112 positions per block -
    // every block streams the 0.4 MB of bf16 weights once per 112 positions instead of once per 32 (the 32-position form re-reads
    // them 784 times at the headline: 316 MB through L2 for 10 GFLOP)
#ifndef HEAD_NT32
    if (a.C <= 384 && a.P % 112 == 0) return launch_head_fwd<3, 112, 8>(a, s);
#endif
    // 32 positions per block: 40 KB of LDS and < 256 registers -> two blocks per CU, one block's tile load under the other's MFMAs
    if (a.C <= 384) return launch_head_fwd<6, 32>(a, s);
    if (a.C <= 768) return launch_head_fwd<12, 32>(a, s);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------- backward
// d hidden_pre = (hidden > 0) * (W2b^T d code), bf16 (B, C, P).  Block = (position tile of 64, image); the product is formed
// TRANSPOSED (rows = positions, columns = hidden channels): an accumulator then holds four consecutive positions of one channel,
// which is 8 contiguous bytes of `hidden` (the ReLU mask) and of the output.

// Round 4: the ReLU mask (`hidden`) and the result go through an LDS image [channel][64 positions] - rows of 128 contiguous bytes
// in memory, read and written by eight lanes of 16 bytes each.  The accumulator layout alone gives every lane 8 bytes of 16
// different channel rows per instruction: 32-byte runs, 1.8 TB/s on this kernel's 91 MB.
// NKS > 0: the weight fragments of all NKS k-steps (D <= 32 NKS) requested with the tile; 0: fetched inside the k-loop
template <int MB, int NKS = 0>
__global__ __launch_bounds__(256) void k_head_dh(const DgHeadDhArgs a) {
    constexpr int NT = 64, FROW = NT * 2 + 32, DPMAX = 128, HROW = NT * 2, CP = 16 * MB * 4;
    extern __shared__ __attribute__((aligned(1024))) char dh_sm[];
    char* const Dt = dh_sm;                                            // [DPMAX][FROW]: [d][position] bf16, zero padded to a multiple of 32 rows
    char* const Ht = dh_sm + DPMAX * FROW;                             // [CP][HROW]: hidden tile in, d hidden tile out (staged form only)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.y, p0 = blockIdx.x * NT;
    const int C = a.C, D = a.D, P = a.P, DP = (D + 31) / 32 * 32;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    const bool staged = a.staged != 0;                                 // (P a multiple of 8: every 16-byte piece of a row is whole)
#ifdef DG_DEVTOOLS
    // two blocks' phase stamps: an early one and one of the second round (blocks 3 and 700 of the 13 x 64 grid)
#define DSTAMP(k) if (a.stamps && tid == 0 && (blk == 3 || blk == 700)) a.stamps[(blk == 3 ? 0 : 8) + k] = __builtin_amdgcn_s_memtime();
#else
#define DSTAMP(k)
#endif
    DSTAMP(0)
    // hidden rows of the tile: all of a thread's 16-byte pieces requested up front, next to the d code pieces below (one round trip)
    // Round 6: by LDS-DMA (no registers: 48 of them held the pieces until the d code tile was converted), rows of 128 bytes, the
    // 16-byte pieces of row m XORed with m & 7 - the epilogue's 8-byte cells of 16 different rows and the row copies are both conflict-free
    constexpr int HPCS = CP * (NT / 8) / 256;                          // pieces per thread (the copy-out)
    if (staged) {
        const uint32_t hdst = lds_addr(Ht);
#pragma unroll
        for (int u = 0; u < CP / 32; ++u) {                            // KiB pieces of 8 rows: wid, wid + 4, ...
            const int k = __builtin_amdgcn_readfirstlane(wid) + 4 * u, m = 8 * k + (lane >> 3), pcl = (lane & 7) ^ (m & 7), p = p0 + 8 * pcl;
            dma16(a.hidden + ((size_t)b * C + (m < C ? m : C - 1)) * P + (p + 7 < P ? p : P - 8), hdst + k * 1024);
        }
    }
    // the weight fragments of every k-step (B[d][channel m] = W2bT[m][d]; C <= 384, D <= 96: up to 3 x 6 16-byte pieces per lane), requested
    // with the tile: inside the k-loop each step waited an L2 round trip of its own (7.7 k cycles for 72 MFMAs, stamps of round 6)
    const int nbase = wid * 16 * MB;
    u32x4 wpre[NKS > 0 ? NKS : 1][NKS > 0 ? MB : 1];
    if constexpr (NKS > 0) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int j = 0; j < MB; ++j) wpre[ks][j] = wraw(a.w2bT, DP, nbase + 16 * j + c16, C, 32 * (ks < DP / 32 ? ks : 0), DP, lane);
    }
    // d code tile -> LDS; its row sums over the tile are the block's share of d b1 (= d b2b): 16 consecutive lanes hold one row
    if ((P & 3) == 0) {
        // all of a thread's pieces requested first (clamped addresses), then used: one memory round trip for the tile instead of
        // one per piece (each piece's row sum was formed right behind its loads)
        constexpr int NP = DPMAX * (NT / 4) / 256;             // 8 pieces per thread at most
        f32x4 v[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int idx = tid + 256 * u, d = idx / (NT / 4), q4 = idx - d * (NT / 4), p = p0 + 4 * q4;
            v[u] = *reinterpret_cast<const f32x4*>(a.gcode + dg_img_off(b, (long long)D * P, a.Bs, a.d_gcode) + (size_t)(d < D ? d : D - 1) * P + (p < P ? p : P - 4));
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int idx = tid + 256 * u, d = idx / (NT / 4), q4 = idx - d * (NT / 4), p = p0 + 4 * q4;
            if (idx < DP * (NT / 4)) {                        // (uniform per wave: 16 consecutive lanes hold one row)
                const bool ok = d < D && p < P;
                bf16x4 o;
                float rs = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float x = ok ? v[u][e] : 0.f; rs += x; o[e] = (__bf16)x; }
                *reinterpret_cast<bf16x4*>(Dt + d * FROW + q4 * 8) = o;
                if (a.gcode_bf && ok) *reinterpret_cast<bf16x4*>(a.gcode_bf + ((size_t)b * D + d) * P + p) = o;     // (for k_head_wgrad3: its d code rows, as rounded here)
#pragma unroll
                for (int sh = 8; sh > 0; sh >>= 1) rs += __shfl_xor(rs, sh, 64);
                if ((tid & 15) == 0 && d < D) a.part_bd[(size_t)blk * D + d] = rs;
            }
        }
    } else {
        for (int idx = tid; idx < DP * (NT / 4); idx += 256) {
            const int d = idx / (NT / 4), q4 = idx - d * (NT / 4), p = p0 + 4 * q4;
            bf16x4 o;
            float rs = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = (d < D && p + e < P) ? a.gcode[dg_img_off(b, (long long)D * P, a.Bs, a.d_gcode) + (size_t)d * P + p + e] : 0.f;
                rs += v;
                o[e] = (__bf16)v;
            }
            *reinterpret_cast<bf16x4*>(Dt + d * FROW + q4 * 8) = o;
#pragma unroll
            for (int sh = 8; sh > 0; sh >>= 1) rs += __shfl_xor(rs, sh, 64);
            if ((tid & 15) == 0 && d < D) a.part_bd[(size_t)blk * D + d] = rs;
        }
    }
    DSTAMP(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (the DMA pieces: hipcc does not count them)
    __syncthreads();
    DSTAMP(2)
    f32x4 acc[4][MB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 ones = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if constexpr (NKS > 0) {
        // (weight fragments: requested with the tile, above)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < DP / 32) {
                bf16x8 af[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = tr_frag(Dt, FROW, 32 * ks, 16 * i, lane);     // A[position][d]: the same transposing read
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wfrag(wpre[ks][j], ones, lane), acc[i][j], 0, 0, 0);
            }
        }
    } else {
    for (int ks = 0; ks < DP / 32; ++ks) {
        bf16x8 af[4], bfr[MB];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = tr_frag(Dt, FROW, 32 * ks, 16 * i, lane);     // A[position][d]: the same transposing read
#pragma unroll
        for (int j = 0; j < MB; ++j)                                                       // B[d][channel m] = W2b[d][m] = W2bT[m][d]
            bfr[j] = wfrag(wraw(a.w2bT, DP, nbase + 16 * j + c16, C, 32 * ks, DP, lane), ones, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    }
    const bool vec = (P & 3) == 0;
    DSTAMP(3)
    if (staged) {
        // mask and result in place in the LDS image (the lane that reads a piece is the one that overwrites it), then whole rows out
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = nbase + 16 * j + c16;
            float bs = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = p0 + 16 * i + 4 * g;
                bf16x4* cell = reinterpret_cast<bf16x4*>(Ht + m * HROW + (((2 * i + (g >> 1)) ^ (m & 7)) << 4) + 8 * (g & 1));
                const bf16x4 h4 = *cell;
                bf16x4 o4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = ((float)h4[r] > 0.f && m < C && p + r < P) ? acc[i][j][r] : 0.f;
                    bs += v;
                    o4[r] = (__bf16)v;
                }
                *cell = o4;
            }
            bs += __shfl_xor(bs, 16, 64);
            bs += __shfl_xor(bs, 32, 64);
            if (g == 0 && m < C) a.part_b2a[(size_t)blk * C + m] = bs;
        }
        __syncthreads();
        DSTAMP(4)
#pragma unroll
        for (int u = 0; u < HPCS; ++u) {
            const int idx = tid + 256 * u, m = idx >> 3, pc = idx & 7, p = p0 + 8 * pc;
            if (m < C && p + 7 < P) *reinterpret_cast<u32x4*>(a.dh + ((size_t)b * C + m) * P + p) = *reinterpret_cast<const u32x4*>(Ht + m * HROW + ((pc ^ (m & 7)) << 4));
        }
        DSTAMP(5)
        return;
    }
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int m = nbase + 16 * j + c16;
        float bs = 0.f;                                     // sum over the tile's positions of d hidden_pre[m]: share of d b2a
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + 16 * i + 4 * g;
            bf16x4 hpre = {};
            if (vec) hpre = *reinterpret_cast<const bf16x4*>(a.hidden + ((size_t)b * C + (m < C ? m : C - 1)) * P + (p < P ? p : P - 4));
            if (m < C && p < P) {
                const size_t off = ((size_t)b * C + m) * P + p;
                bf16x4 h4, o4;
                if (vec) h4 = hpre;
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h4[r] = p + r < P ? a.hidden[off + r] : (__bf16)0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = ((float)h4[r] > 0.f && p + r < P) ? acc[i][j][r] : 0.f;
                    bs += v;
                    o4[r] = (__bf16)v;
                }
                if (vec) *reinterpret_cast<bf16x4*>(a.dh + off) = o4;
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (p + r < P) a.dh[off + r] = o4[r];
                }
            }
        }
        bs += __shfl_xor(bs, 16, 64);
        bs += __shfl_xor(bs, 32, 64);
        if (g == 0 && m < C) a.part_b2a[(size_t)blk * C + m] = bs;
    }
}

// Round 6, the headline widths (192 < C <= 384, D <= 96, P a multiple of 8): persistent blocks, the next tile's loads under the
// current tile's work.  k_head_dh is load -> multiply -> mask -> store per block with two blocks per CU to overlap them: its load
// phase alone is 10-15 us of a block's 18-25 (stamps, experiments/r06.md).  Here a block of eight waves walks tiles bid, bid + grid, ...:
// at the top of tile t the hidden rows of tile t + 1 go global -> LDS by DMA into the second image and its d code pieces into
// registers; the weight fragments (36 registers at 48 channels per wave) are loaded once per block.
#define DH2_HT (384 * 128)
#define DH2_DT (96 * 160)
// ND: 16-row blocks of d code in the fused d W2b product (5: D <= 80, 6: D <= 96; 72 accumulator registers at 6 leave the kernel
// four registers short), 0: not formed
template <int ND>
__global__ __launch_bounds__(512) void k_head_dh2(const DgHeadDhArgs a) {
    constexpr bool W2B = ND > 0;
    constexpr int NT = 64, FROW = NT * 2 + 32, HROW = NT * 2, MB = 3, NKS = 3;
    extern __shared__ __attribute__((aligned(1024))) char dh2_sm[];    // [2][hidden tile 384 x 128 B] [2][d code tile 96 x 160 B]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
    const int C = a.C, D = a.D, P = a.P, DP = (D + 31) / 32 * 32;
    const int tiles_img = (P + NT - 1) / NT, ntiles = a.B * tiles_img, steps_img = (P + 31) / 32;
    const uint32_t ht0 = lds_addr(dh2_sm);
    // the weight fragments of every k-step: B[d][channel m] = W2bT[m][d], wave `wid` owns channels [48 wid, 48 wid + 48)
    const int nbase = wid * 16 * MB;
    u32x4 wpre[NKS][MB];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int j = 0; j < MB; ++j) wpre[ks][j] = wraw(a.w2bT, DP, nbase + 16 * j + c16, C, 32 * (ks < DP / 32 ? ks : 0), DP, lane);
    const u32x4 ones = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    constexpr int NP = 96 * (NT / 4) / 512;                // d code pieces (4 positions, fp32) per thread: 3
    f32x4 v[NP];
    auto fetch = [&](const int t, const int buf) __attribute__((always_inline)) {
        const int b = t / tiles_img, p0 = (t - b * tiles_img) * NT;
        // hidden rows: KiB pieces of 8 rows (wid, wid + 8, ...: six per wave), the 16-byte pieces of row m XORed with m & 7
        const __bf16* hb = a.hidden + (size_t)b * C * P;               // (scalar base + one 32-bit offset per lane and piece)
        const int m0 = 8 * wid + (lane >> 3), pl = p0 + 8 * ((lane & 7) ^ (m0 & 7)), pc = pl + 7 < P ? pl : P - 8;    // (m & 7 is the same for every piece)
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int m = m0 + 64 * u;
            dma16_s(hb, (uint32_t)((m < C ? m : C - 1) * P + pc) * 2, ht0 + buf * DH2_HT + (wid + 8 * u) * 1024);
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int idx = tid + 512 * u, d = idx / (NT / 4), q4 = idx - d * (NT / 4), p = p0 + 4 * q4;
            v[u] = *reinterpret_cast<const f32x4*>(a.gcode + dg_img_off(b, (long long)D * P, a.Bs, a.d_gcode) + (size_t)(d < D ? d : D - 1) * P + (p < P ? p : P - 4));
        }
    };
    // d W2b[d][m] = sum over positions of d code[d][p] hidden[m][p] for the wave's 48 channels m: both tiles are in LDS anyway (the
    // product had a launch of its own, k_head_wgrad2 beside this kernel on the second stream, re-reading both tensors); accumulated
    // over the block's tiles, one partial sum per block
    f32x4 wacc[W2B ? ND : 1][MB];
#pragma unroll
    for (int i = 0; i < (W2B ? ND : 1); ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) wacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (rows DP .. 95 of the d code images are never written: zero, the d W2b product reads all 96)
    for (int i = tid; i < 2 * DH2_DT / 16; i += 512) {
        const int bufi = i / (DH2_DT / 16), o = i - bufi * (DH2_DT / 16);
        if (o * 16 >= DP * FROW) *reinterpret_cast<u32x4*>(dh2_sm + 2 * DH2_HT + bufi * DH2_DT + o * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    int t = blockIdx.x, buf = 0;
#ifdef DG_DEVTOOLS
    int tix = 0;
#define D2STAMP(k) if (a.stamps && tid == 0 && blockIdx.x == 3 && tix < 2) a.stamps[8 * tix + k] = __builtin_amdgcn_s_memtime();
#else
#define D2STAMP(k)
#endif
    if (t < ntiles) fetch(t, 0);
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        const int b = t / tiles_img, p0 = (t - b * tiles_img) * NT;
        D2STAMP(0)
        char* const Ht = dh2_sm + buf * DH2_HT;
        char* const Dt = dh2_sm + 2 * DH2_HT + buf * DH2_DT;
        // d code tile -> LDS (bf16), its row sums = the tile's share of d b1 (= d b2b), the bf16 copy for k_head_wgrad3
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int idx = tid + 512 * u, d = idx / (NT / 4), q4 = idx - d * (NT / 4), p = p0 + 4 * q4;
            if (d < DP) {                                     // (uniform per wave: 16 consecutive lanes hold one row)
                const bool ok = d < D && p < P;
                bf16x4 o;
                float rs = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float x = ok ? v[u][e] : 0.f; rs += x; o[e] = (__bf16)x; }
                *reinterpret_cast<bf16x4*>(Dt + d * FROW + q4 * 8) = o;
                if (a.gcode_bf && ok) {
                    if (a.step_major) *reinterpret_cast<bf16x4*>(a.gcode_bf + (((size_t)b * steps_img + (p >> 5)) * D + d) * 32 + (p & 31)) = o;
                    else *reinterpret_cast<bf16x4*>(a.gcode_bf + ((size_t)b * D + d) * P + p) = o;
                }
#pragma unroll
                for (int sh = 8; sh > 0; sh >>= 1) rs += __shfl_xor(rs, sh, 64);
                if ((tid & 15) == 0 && d < D) a.part_bd[(size_t)t * D + d] = rs;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the DMA pieces of this tile: hipcc does not count them)
        __syncthreads();                                               // tile t in LDS; every wave is done with the other images (tile t - 1)
        D2STAMP(1)
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x, buf ^ 1);
        f32x4 acc[4][MB];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < DP / 32) {
                bf16x8 af[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = tr_frag(Dt, FROW, 32 * ks, 16 * i, lane);     // A[position][d]
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wfrag(wpre[ks][j], ones, lane), acc[i][j], 0, 0, 0);
            }
        }
        D2STAMP(2)
        if constexpr (W2B) {
            // A[d][k = position] rows of the d code image, B[k][m] = hidden[m][k] rows of the hidden image (before the mask pass below
            // overwrites them): 16-byte reads, natural k order on both sides.  Positions beyond the image: the d code image holds zeros
#pragma unroll
            for (int ks = 0; ks < NT / 32; ++ks) {
                bf16x8 hb[MB];
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    const int m = nbase + 16 * j + c16;
                    hb[j] = *reinterpret_cast<const bf16x8*>(Ht + m * HROW + (((4 * ks + g) ^ (m & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const bf16x8 ga = *reinterpret_cast<const bf16x8*>(Dt + (16 * i + c16) * FROW + (32 * ks + 8 * g) * 2);
#pragma unroll
                    for (int j = 0; j < MB; ++j) wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga, hb[j], wacc[i][j], 0, 0, 0);
                }
            }
        }
        D2STAMP(3)
        // mask and result in place in the LDS image (the lane that reads a cell is the one that overwrites it); row sums of the result
        float bs[MB];
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = nbase + 16 * j + c16;
            bs[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = p0 + 16 * i + 4 * g;
                bf16x4* cell = reinterpret_cast<bf16x4*>(Ht + m * HROW + (((2 * i + (g >> 1)) ^ (m & 7)) << 4) + 8 * (g & 1));
                const bf16x4 h4 = *cell;
                bf16x4 o4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = ((float)h4[r] > 0.f && m < C && p + r < P) ? acc[i][j][r] : 0.f;
                    bs[j] += x;
                    o4[r] = (__bf16)x;
                }
                *cell = o4;
            }
        }
#pragma unroll
        for (int j = 0; j < MB; ++j) bs[j] += __shfl_xor(bs[j], 16, 64);
#pragma unroll
        for (int j = 0; j < MB; ++j) bs[j] += __shfl_xor(bs[j], 32, 64);
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = nbase + 16 * j + c16;
            if (g == 0 && m < C) a.part_b2a[(size_t)t * C + m] = bs[j];
        }
        __syncthreads();
        D2STAMP(4)
        // whole rows out: 384 rows x 8 pieces of 16 bytes = six per thread
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int idx = tid + 512 * u, m = idx >> 3, pc = idx & 7, p = p0 + 8 * pc;
            const u32x4 val = *reinterpret_cast<const u32x4*>(Ht + m * HROW + ((pc ^ (m & 7)) << 4));
            if (a.step_major) {
                // [image][step][row][32 positions]: the tile is steps 2 tile, 2 tile + 1; positions beyond the image hold zeros (masked above)
                const int stp = (p0 >> 5) + (pc >> 2);
                if (m < C && stp < steps_img) *reinterpret_cast<u32x4*>(a.dh + (((size_t)b * steps_img + stp) * C + m) * 32 + 8 * (pc & 3)) = val;
            } else if (m < C && p + 7 < P) *reinterpret_cast<u32x4*>(a.dh + ((size_t)b * C + m) * P + p) = val;
        }
        D2STAMP(5)
#ifdef DG_DEVTOOLS
        ++tix;
#endif
    }
    if constexpr (W2B) {
        float* out = a.part_w2b + (size_t)blockIdx.x * D * C;
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const int m = nbase + 16 * j + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * i + 4 * g + r;
                    if (d < D && m < C) out[(size_t)d * C + m] = wacc[i][j][r];
                }
            }
    }
}

int dg_head_dh_fused_blocks(int B, int C, int D, int P) {
#ifdef DG_DEVTOOLS
    if (const char* e = getenv("DG_HEAD_DH2")) if (e[0] == '0') return 0;
#endif
    if (!(C > 192 && C <= 384 && D <= 96 && (P & 7) == 0)) return 0;
    const int ntiles = B * ((P + 63) / 64);
    return ntiles < 256 ? ntiles : 256;
}
hipError_t dg_launch_head_dh(const DgHeadDhArgs& a, hipStream_t s) {
    dim3 grid((a.P + 63) / 64, a.B);
    DgHeadDhArgs a2 = a;
    if (const int nblk = dg_head_dh_fused_blocks(a.B, a.C, a.D, a.P)) {
        const int smem = 2 * DH2_HT + 2 * DH2_DT;
        auto kern = !a.part_w2b ? k_head_dh2<0> : (a.D <= 80 ? k_head_dh2<5> : k_head_dh2<6>);
        hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
        if (e != hipSuccess) return e;
#ifdef DG_DEVTOOLS
        if (const char* sf = getenv("DG_DH_STAMPS")) {
            static unsigned long long* sb2 = nullptr;
            if (!sb2 && hipMalloc(&sb2, 128) != hipSuccess) return hipErrorOutOfMemory;
            a2.stamps = sb2;
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), smem, s, a2);
            unsigned long long hs[16];
            if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hs, sb2, 128, hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE* fp = fopen(sf, "w")) { for (int q = 0; q < 2; ++q) for (int i = 1; i < 6; ++i)
                    fprintf(fp, "k_head_dh2 block 3 tile %d phase %d: %llu cycles\n", q, i, hs[8 * q + i] - hs[8 * q + i - 1]);
                    fprintf(fp, "tile 0 start -> tile 1 start: %llu cycles\n", hs[8] - hs[0]); fclose(fp); }
            return hipGetLastError();
        }
#endif
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), smem, s, a2);
        return hipGetLastError();
    }
    if (a.part_w2b) return hipErrorInvalidValue;           // (only k_head_dh2 forms d W2b: the plan asks dg_head_dh_fused_blocks first)
#ifdef DG_DEVTOOLS
    static unsigned long long* sbuf = nullptr;
    const char* sfile = getenv("DG_DH_STAMPS");
    if (sfile) { if (!sbuf && hipMalloc(&sbuf, 128) != hipSuccess) return hipErrorOutOfMemory; a2.stamps = sbuf; }
#endif
    a2.staged = ((a.P & 7) == 0 && a.C <= 384) ? 1 : 0;     // (ViT-B width: 130 KB of LDS would leave one block per CU)
#ifdef DG_DEVTOOLS
#define DH_STAMP_DUMP if (sfile) { unsigned long long hs[16];                                                    \
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hs, sbuf, 128, hipMemcpyDeviceToHost) == hipSuccess)     \
            if (FILE* fp = fopen(sfile, "w")) { for (int q = 0; q < 2; ++q) for (int i = 1; i < 6; ++i)          \
                fprintf(fp, "block %d phase %d: %llu cycles (starts %llu after block 3)\n", q ? 700 : 3, i, hs[8 * q + i] - hs[8 * q + i - 1], hs[8 * q + i - 1] - hs[0]); fclose(fp); } }
#else
#define DH_STAMP_DUMP
#endif
#define DG_DH(MB_, NKS_) {                                                                                       \
        const int smem = 128 * (64 * 2 + 32) + (a2.staged ? 16 * MB_ * 4 * (64 * 2) : 0);                        \
        hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_head_dh<MB_, NKS_>), smem);               \
        if (e != hipSuccess) return e;                                                                           \
        hipLaunchKernelGGL((k_head_dh<MB_, NKS_>), grid, dim3(256), smem, s, a2);                                \
        DH_STAMP_DUMP                                                                                            \
        return hipGetLastError(); }
    const bool pre3 = a.D <= 96;        // (all weight fragments in registers: 72 of them at C = 384; a fourth k-step would leave one wave per SIMD)
    if (a.C <= 64) { if (pre3) DG_DH(1, 3) else DG_DH(1, 4) }
    if (a.C <= 128) { if (pre3) DG_DH(2, 3) else DG_DH(2, 4) }
    if (a.C <= 192) { if (pre3) DG_DH(3, 3) else DG_DH(3, 4) }
    if (a.C <= 384) { if (pre3) DG_DH(6, 3) else DG_DH(6, 0) }
    if (a.C <= 768) DG_DH(12, 0)
#undef DG_DH
    return hipErrorInvalidValue;
}

// Weight gradient: part[split][m][n] = sum over the split's (image, 32-position steps) of keep[image][n] * A[image][m][p] * Bm[image][n][p].
// Both operands have the position index contiguous, so fragments are plain 16 / 32-byte global loads (no LDS).  Block = 4 waves
// as 2 x 2, each wave a 64 x 64 output tile (16 accumulators); blockIdx.z = split of the position steps.

template <typename T>
__device__ __forceinline__ bf16x8 row_frag(const T* __restrict__ X, const int row, const int nrows, const int P, const int p, const bool vec) {
    bf16x8 o;
    if (row < nrows && vec && p + 7 < P) {
        if constexpr (sizeof(T) == 2) return *reinterpret_cast<const bf16x8*>(X + (size_t)row * P + p);
        else {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(X + (size_t)row * P + p), hi = *reinterpret_cast<const f32x4*>(X + (size_t)row * P + p + 4);
            return pack8(lo, hi);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (row < nrows && p + e < P) ? (__bf16)(float)X[(size_t)row * P + p + e] : (__bf16)0.f;
    return o;
}

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void k_head_wgrad(const DgHeadWgradArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int m0 = blockIdx.y * 128 + (wid >> 1) * 64, n0 = blockIdx.x * 128 + (wid & 1) * 64;
    const int steps_img = (a.P + 31) / 32, total = a.B * steps_img;
    const int s0 = (int)((long long)total * blockIdx.z / a.splits), s1 = (int)((long long)total * (blockIdx.z + 1) / a.splits);
    const bool vec = (a.P & 7) == 0;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool live = m0 < a.M && n0 < a.N;
    // the fragments of step s + 1 are loaded (raw) while step s multiplies: every step would otherwise wait a full memory latency
    using RA = typename std::conditional<sizeof(TA) == 2, u32x4, f32x4>::type;
    using RB = typename std::conditional<sizeof(TB) == 2, u32x4, f32x4>::type;
    constexpr int WA = sizeof(TA) == 2 ? 1 : 2, WB = sizeof(TB) == 2 ? 1 : 2;
    RA ra[4][WA];
    RB rb[4][WB];
    float kp[4];
    auto load_raw = [&](auto& dst, const auto* X, const int row, const int nrows, const int p) {
        using T = typename std::remove_cv<typename std::remove_pointer<decltype(X)>::type>::type;
        constexpr int W = sizeof(T) == 2 ? 1 : 2;
        const bool ok = row < nrows && p + 7 < a.P;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (ok && vec) dst[w] = *reinterpret_cast<const typename std::remove_reference<decltype(dst[0])>::type*>(X + (size_t)row * a.P + p + 4 * w);
            else {
                if constexpr (sizeof(T) == 2) {
                    bf16x8 t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (row < nrows && p + e < a.P) ? X[(size_t)row * a.P + p + e] : (__bf16)0.f;
                    dst[w] = __builtin_bit_cast(u32x4, t);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[w][e] = (row < nrows && p + 4 * w + e < a.P) ? X[(size_t)row * a.P + p + 4 * w + e] : 0.f;
                }
            }
        }
    };
    auto fetch = [&](const int s) {
        const int b = s / steps_img, p = (s - b * steps_img) * 32 + 8 * g;
        const TA* Ab = static_cast<const TA*>(a.A) + dg_img_off(b, (long long)a.M * a.P, a.Bs, a.dA);
        const TB* Bb = static_cast<const TB*>(a.Bm) + dg_img_off(b, (long long)a.N * a.P, a.Bs, a.dB);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_raw(ra[i], Ab, m0 + 16 * i + c16, a.M, p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 16 * j + c16;
            load_raw(rb[j], Bb, n, a.N, p);
            kp[j] = (a.keep && n < a.N) ? a.keep[(size_t)b * a.N + n] : 1.f;
        }
    };
    auto cook = [&](const auto& r) {
        if constexpr (sizeof(r[0]) == 16 && std::is_same<typename std::remove_cv<typename std::remove_reference<decltype(r[0])>::type>::type, u32x4>::value)
            return __builtin_bit_cast(bf16x8, r[0]);
        else return pack8(r[0], r[1]);
    };
    if (live && s0 < s1) fetch(s0);
    for (int s = s0; s < s1 && live; ++s) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = cook(ra[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { bfr[j] = cook(rb[j]); if (kp[j] == 0.f) bfr[j] = bf16x8{}; }
        if (s + 1 < s1) fetch(s + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (!live) return;
    float* out = a.part + (size_t)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 16 * j + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * i + 4 * g + r;
                if (m < a.M && n < a.N) out[(size_t)m * a.N + n] = acc[i][j][r];
            }
        }
}

// The same product staged through LDS (P a multiple of 8: every 16-byte piece of a row is whole): block = 128 x 128 output tile,
// k-step = 32 positions.  Two threads per operand row load its 64 (bf16) / 128 (fp32) contiguous bytes of the step -
// whole lines, where the direct form above reads 16 bytes per lane from 64 different rows per instruction and is bound by the
// address path -, convert to bf16 and store [row][32 positions] images (80-byte rows: eight consecutive lanes of a 16-byte
// fragment read fall into eight different bank groups); each wave owns a 64 x 64 sub-tile = 2 x 2 MFMAs of 32x32x16 per 16
// positions.  The loads of step s + 1 are in flight while step s multiplies; one barrier per step.
template <typename TA, typename TB>
__device__ __forceinline__ void wgrad2_body(const DgHeadWgradArgs& a, const void* Aop, const long long dAop, const float* keep, float* part,
                                            const int M, const int m0, const int n0, const int split) {
    constexpr int PS = 32, RS = PS * 2 + 16, TILE = 128 * RS;     // positions per step; row stride (80 bytes)
    extern __shared__ __attribute__((aligned(16))) char wsm[];     // [2 stages][A, B][TILE]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int steps_img = (a.P + PS - 1) / PS, total = a.B * steps_img;
    const int s0 = (int)((long long)total * split / a.splits), s1 = (int)((long long)total * (split + 1) / a.splits);
    // loader: two threads per operand row, each with 16 consecutive positions of the step.  (Dealing the 16-byte pieces to the
    // threads in row-major order - 8 / 16 lines per wave instruction instead of 32 - was slower: 36 / 27 / 17 -> 40 / 39 / 15 us for
    // the three products, its per-piece index arithmetic costs 40 registers; 64 positions per step: 50 / 30 / 20 us.)
    constexpr int WA = sizeof(TA) == 2 ? 2 : 4, WB = sizeof(TB) == 2 ? 2 : 4;      // 16-byte loads per thread and step
    using RA = typename std::conditional<sizeof(TA) == 2, u32x4, f32x4>::type;
    using RB = typename std::conditional<sizeof(TB) == 2, u32x4, f32x4>::type;
    RA ra[WA];
    RB rb[WB];
    bool oka[2], okb[2];                  // the two 8-position chunks of this thread: inside the matrix and the image
    const int lrow = tid >> 1, half = tid & 1;
    auto fetch = [&](const int s) {
        const int b = s / steps_img, p = (s - b * steps_img) * PS + 16 * half;
        const int m = m0 + lrow, n = n0 + lrow;
        const TA* Ar = static_cast<const TA*>(Aop) + dg_img_off(b, (long long)M * a.P, a.Bs, dAop) + (size_t)(m < M ? m : 0) * a.P;
        const TB* Br = static_cast<const TB*>(a.Bm) + dg_img_off(b, (long long)a.N * a.P, a.Bs, a.dB) + (size_t)(n < a.N ? n : 0) * a.P;
        const float kp = (keep && n < a.N) ? keep[(size_t)b * a.N + n] : 1.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bool in = p + 8 * c + 7 < a.P;
            const int pc = in ? p + 8 * c : 0;                     // (always a valid address; the value is selected away)
            oka[c] = in && m < M;
            okb[c] = in && n < a.N && kp != 0.f;
            if constexpr (sizeof(TA) == 2) ra[c] = *reinterpret_cast<const u32x4*>(Ar + pc);
            else { ra[2 * c] = *reinterpret_cast<const f32x4*>(Ar + pc); ra[2 * c + 1] = *reinterpret_cast<const f32x4*>(Ar + pc + 4); }
            if constexpr (sizeof(TB) == 2) rb[c] = *reinterpret_cast<const u32x4*>(Br + pc);
            else { rb[2 * c] = *reinterpret_cast<const f32x4*>(Br + pc); rb[2 * c + 1] = *reinterpret_cast<const f32x4*>(Br + pc + 4); }
        }
    };
    auto stash = [&](const int buf) {
        char* At = wsm + buf * 2 * TILE + lrow * RS + half * 32;
        char* Bt = At + TILE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            u32x4 va, vb;
            if constexpr (sizeof(TA) == 2) va = ra[c]; else va = __builtin_bit_cast(u32x4, pack8(ra[2 * c], ra[2 * c + 1]));
            if constexpr (sizeof(TB) == 2) vb = rb[c]; else vb = __builtin_bit_cast(u32x4, pack8(rb[2 * c], rb[2 * c + 1]));
            *reinterpret_cast<u32x4*>(At + 16 * c) = oka[c] ? va : u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(Bt + 16 * c) = okb[c] ? vb : u32x4{0u, 0u, 0u, 0u};
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int wm = wid >> 1, wn = wid & 1, r32 = lane & 31, kg = lane >> 5;
    if (s0 < s1) { fetch(s0); stash(0); }
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        if (s + 1 < s1) fetch(s + 1);
        const char* At = wsm + buf * 2 * TILE + (wm * 64 + r32) * RS + kg * 16;
        const char* Bt = wsm + buf * 2 * TILE + TILE + (wn * 64 + r32) * RS + kg * 16;
        constexpr int KSN = PS / 16;
        bf16x8 af[2][KSN], bfr[2][KSN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                af[i][ks] = *reinterpret_cast<const bf16x8*>(At + i * 32 * RS + ks * 32);
                bfr[i][ks] = *reinterpret_cast<const bf16x8*>(Bt + i * 32 * RS + ks * 32);
            }
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ks], bfr[j][ks], acc[i][j], 0, 0, 0);
        if (s + 1 < s1) stash(buf ^ 1);
        __syncthreads();
    }
    float* out = part + (size_t)split * M * a.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
                if (m < M && n < a.N) out[(size_t)m * a.N + n] = acc[i][j][e];
            }
        }
}

// One launch, one or two products that share the operand Bm (the weight gradients of cluster1 and of cluster2's first convolution
// both multiply the feature tile): tiles of A (M rows) first, then tiles of A2 (M2 rows, its own keep mask and output).
// XCD-aware order: blocks are dealt round-robin over the 8 XCDs, each with its own L2.  All output tiles of one split read the same
// position range of the operands: they get consecutive slots on ONE XCD, so the range comes from HBM once and its re-reads are
// L2 hits (50 -> 36 us for the 384 x 384 product).  (splits is a multiple of 8; grid = tiles x splits, one dimension)
template <typename TA, typename TB, typename TA2>
__global__ __launch_bounds__(256, 2) void k_head_wgrad2(const DgHeadWgradArgs a) {
    const int tn = (a.N + 127) / 128, tm1 = (a.M + 127) / 128, tm2 = (a.M2 + 127) / 128, ntile = tn * (tm1 + tm2);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = xcd + 8 * (slot / ntile), tile = slot % ntile;
    const int tm = tile / tn, n0 = (tile - tm * tn) * 128;
    if (tm < tm1) wgrad2_body<TA, TB>(a, a.A, a.dA, a.keep, a.part, a.M, tm * 128, n0, split);
    else wgrad2_body<TA2, TB>(a, a.A2, a.dA2, a.keep_2, a.part2, a.M2, (tm - tm1) * 128, n0, split);
}

// Round 6: the two products that share the feature operand in ONE pass over it.  A block owns ALL rows of both products - 384 rows of
// d hidden and up to 128 rows of d code (bf16 both: k_head_dh leaves a bf16 copy of d code, the rounding this product applied anyway) =
// a 512-row tile - and 128 feature channels; its eight waves sit 4 x 2 on the 512 x 128 tile (128 x 64 each: 128 accumulator
// registers).  The fp32 feature rows of a step are read ONCE per block, where k_head_wgrad2 reads them once per 128-row tile: four
// times (308 of its 451 MB at the paired headline shape, 77 here).
// One block per CU: nothing but the block's own look-ahead covers the memory round trip, and registers cannot (two sets of a step's
// pieces spill).  So every operand goes global -> LDS by DMA (global_load_lds_dwordx4: no registers), two steps ahead, as it lies in
// memory - the fp32 feature rows stay fp32 in LDS and are rounded to bf16 on the way into the fragments (the same rounding, later).
// LDS image of a step (32 positions): rows of 64 (bf16) / 128 (fp32) bytes, the 16-byte pieces of a row XORed with row bits so that the
// lane groups of a ds_read_b128 cover all 64 banks; a DMA piece = 1 KiB = 16 / 8 consecutive rows.
// Ragged steps (the last of an image) and rows beyond the matrix: the piece is fetched from a clamped, valid address (finite values)
// and the feature fragment is zeroed in registers - as are the channels Dropout2d removed (per image and channel = per lane); rows of
// the tile beyond M2 are never filled: they feed accumulator rows that are never stored.
// grid = channel tiles x splits, the tiles of a split on one XCD (they share the 512 rows).
#define W3_A_BYTES (512 * 64)
#define W3_B_BYTES (128 * 128)
#define W3_STAGE (W3_A_BYTES + W3_B_BYTES)
#define W3_NSTAGE 3
__device__ __forceinline__ void w3_wait_barrier(const int n) {
    switch (n) {
#define W3_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        W3_W(1) W3_W(2) W3_W(3) W3_W(4) W3_W(5) W3_W(6) W3_W(7)
#undef W3_W
        default: asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;    // (over-waits, never under-waits)
    }
}
__global__ __launch_bounds__(512) void k_head_wgrad3(const DgHeadWgradArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char w3sm[];     // [3 stages][A: d hidden 384 rows, d code 128 rows | features 128 rows]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = (a.N + 127) / 128;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = xcd + 8 * (slot / tn), n0 = (slot % tn) * 128;
    const int steps_img = (a.P + 31) / 32, total = a.B * steps_img;
    const int s0 = (int)((long long)total * split / a.splits), s1 = (int)((long long)total * (split + 1) / a.splits), ns = s1 - s0;
    const int wm = wid >> 1, wn = wid & 1, r32 = lane & 31, kg = lane >> 5;
    const int g16 = (a.M2 + 15) >> 4;                                 // 1-KiB pieces of the d code rows
    const uint32_t lds0 = lds_addr(w3sm);
    // ---- the wave's DMA pieces.  A lane's 16 bytes: LDS slot (row, physical piece) = the lane's place in the KiB; it fetches the LOGICAL
    //      piece physical ^ swizzle(row).  A rows (64 bytes): pieces wid, wid + 8, wid + 16 of d hidden, piece wid of d code (below g16);
    //      feature rows (128 bytes): pieces wid, wid + 8.  Per lane: the byte offset of its row inside an image (one register per piece)
    //      and of its logical piece inside the row; per step: a scalar base (image, first position) per tensor
    //      d hidden: row = 16 piece + lane / 4, swizzle (row >> 2) & 3 = (lane >> 4) & 3 for every piece
    const int pos_a = 8 * ((lane & 3) ^ ((lane >> 4) & 3));          // first position of the lane's piece (bf16 rows)
    //      fp32 rows: row = 8 piece + lane / 8, swizzle (row >> 1) & 7 = (4 (piece & 1) + (lane >> 4)) & 7, piece & 1 = wid & 1
    const int pos_f = 4 * ((lane & 7) ^ ((4 * (wid & 1) + (lane >> 4)) & 7));
    uint32_t ro_dh[3], ro_g, ro_f[2];
    const int arow = a.a_step_major ? 64 : a.P * 2;                    // bytes between two rows of A inside a step
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int row = 16 * (wid + 8 * u) + (lane >> 2); ro_dh[u] = (uint32_t)(row < a.M ? row : a.M - 1) * arow; }
    { const int row = 16 * wid + (lane >> 2); ro_g = (uint32_t)(row < a.M2 ? row : a.M2 - 1) * arow; }
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int row = n0 + 8 * (wid + 8 * u) + (lane >> 3); ro_f[u] = (uint32_t)(row < a.N ? row : a.N - 1) * a.P * 4; }
    const bool has_g = wid < g16;
    const int npw = 5 + (has_g ? 1 : 0);                               // DMA instructions of this wave per step
    auto issue = [&](int b, int p0, const int st) __attribute__((always_inline)) {
#ifdef W3_ABL_NODMA
        if (b >= 0) return;                              // (developer ablation, WRONG results: nothing is fetched)
#endif
#ifdef W3_ABL_SAMESTEP
        b = s0 / steps_img; p0 = 32 * (st & 1);          // (developer ablation, WRONG results: the block re-reads two steps - every piece an L2 hit)
#endif
        const int v = a.P - p0;                                        // valid positions from p0 on (>= 8)
        const uint32_t dst = lds0 + st * W3_STAGE + wid * 1024;
        const uint32_t pa = (pos_a + 8 <= v ? pos_a : 0) * 2, pf = (pos_f + 4 <= v ? pos_f : 0) * 4;
        // (step-major A: [image][step][row][32 positions] - a piece = 16 rows x 64 bytes = one contiguous KiB)
        const size_t sidx = (size_t)b * steps_img + (p0 >> 5);
        const __bf16* Ab = a.a_step_major ? static_cast<const __bf16*>(a.A) + sidx * a.M * 32
                                          : static_cast<const __bf16*>(a.A) + dg_img_off(b, (long long)a.M * a.P, a.Bs, a.dA) + p0;
#pragma unroll
        for (int u = 0; u < 3; ++u) dma16_s(Ab, ro_dh[u] + pa, dst + u * 8192);
        if (has_g) dma16_s(a.a_step_major ? static_cast<const __bf16*>(a.A2h) + sidx * a.M2 * 32 : static_cast<const __bf16*>(a.A2h) + (size_t)b * a.M2 * a.P + p0,
                           ro_g + pa, dst + 24 * 1024);
        const float* Bb = static_cast<const float*>(a.Bm) + dg_img_off(b, (long long)a.N * a.P, a.Bs, a.dB) + p0;
#pragma unroll
        for (int u = 0; u < 2; ++u) dma16_s(Bb, ro_f[u] + pf, dst + W3_A_BYTES + u * 8192);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // fragment addresses inside a stage (bytes): row and swizzle are fixed per lane, the wave's row blocks lie 32 rows apart (an
    // immediate offset)
    const int asw = (r32 >> 2) & 3, a_l = (wm * 128 + r32) * 64;
    const int a_off[2] = {a_l + ((kg ^ asw) << 4), a_l + (((2 + kg) ^ asw) << 4)};
    const int bsw = (r32 >> 1) & 7, b_l = W3_A_BYTES + (wn * 64 + r32) * 128;
    const float* const kmask = wm == 3 ? a.keep_2 : a.keep;
    unsigned kbits[2] = {~0u, ~0u};         // all-ones / zero per feature channel of this lane: Dropout2d and channels beyond N
    int bcur = -1;
    auto load_b = [&](const char* base, const int ks, const int v, bf16x8 (&bfr)[2]) __attribute__((always_inline)) {
        const int L0 = 4 * ks + 2 * kg;
        const unsigned live = (16 * ks + 8 * kg + 8 > v) ? 0u : ~0u;   // positions beyond the image
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(base + b_l + j * 4096 + ((L0 ^ bsw) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(base + b_l + j * 4096 + (((L0 + 1) ^ bsw) << 4));
            u32x4 t = __builtin_bit_cast(u32x4, pack8(lo, hi));
            t &= (kbits[j] & live);
            bfr[j] = __builtin_bit_cast(bf16x8, t);
        }
    };
    auto load_a = [&](const char* base, const int ks, bf16x8 (&af)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + a_off[ks] + i * 2048);
    };
    constexpr int NSTAGE = W3_NSTAGE, DIST = NSTAGE - 1;
    // (image, first position) of the step being issued and of the step being multiplied: advanced, not divided
    int bi = s0 / steps_img, pi = (s0 - bi * steps_img) * 32, bm = bi, pm = pi;
#pragma unroll
    for (int k = 0; k < DIST; ++k)
        if (k < ns) { issue(bi, pi, k); pi += 32; if (pi >= a.P) { pi = 0; ++bi; } }
    int st = 0, stn = DIST % NSTAGE;        // stage of step k; stage step k + DIST goes to
    const bool late = wid >= 4;
    bf16x8 b0[2] = {}, b1[2] = {}, a0[4] = {}, a1[4] = {};
    for (int k = 0; k < ns; ++k) {
        const int newer = ns - 1 - k < DIST - 1 ? ns - 1 - k : DIST - 1;       // steps issued behind step k
        w3_wait_barrier(newer * npw);        // step k has landed (every wave's pieces) and nobody reads the stage step k + DIST overwrites
        if (k + DIST < ns) { issue(bi, pi, stn); pi += 32; if (pi >= a.P) { pi = 0; ++bi; } }
        if (bm != bcur) {
            bcur = bm;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nj = n0 + wn * 64 + j * 32 + r32;
                const float kv = (kmask && nj < a.N) ? kmask[(size_t)bm * a.N + nj] : 1.f;
                kbits[j] = (nj < a.N && kv != 0.f) ? ~0u : 0u;
            }
        }
        const int v = a.P - pm;
        const char* base = w3sm + st * W3_STAGE;
        // Waves 0-3 read the step's fragments and multiply; waves 4-7 (the SECOND wave of each SIMD) multiply the fragments they read in
        // the previous step and then read this step's: behind the barrier one wave of a SIMD is on the LDS pipe while the other is on the
        // matrix core.  The MFMAs stay unconditional (a condition around them makes hipcc shuffle the accumulators): the late waves'
        // first product is on zeros
#ifdef W3_ABL_NOMUL
        if (v == -12345)                                 // (developer ablation, WRONG results: no fragment reads, no products)
#endif
        {
        if (!late) { load_b(base, 0, v, b0); load_b(base, 1, v, b1); load_a(base, 0, a0); load_a(base, 1, a1); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        if (late) { load_b(base, 0, v, b0); load_b(base, 1, v, b1); load_a(base, 0, a0); load_a(base, 1, a1); }
        }
        pm += 32; if (pm >= a.P) { pm = 0; ++bm; }
        st = st + 1 == NSTAGE ? 0 : st + 1;
        stn = stn + 1 == NSTAGE ? 0 : stn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (no DMA piece outlives the block's LDS; all have landed by the last step anyway)
    // the late waves' last step (the others: once more on zeros)
    if (!late) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { b0[j] = bf16x8{}; b1[j] = bf16x8{}; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
    const int Mo = wm == 3 ? a.M2 : a.M;
    float* out = (wm == 3 ? a.part2 : a.part) + (size_t)split * Mo * a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (wm == 3 ? 0 : wm * 128) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
#ifdef W3_ABL_NOSTORE
                if (acc[i][j][e] == 1.2345f)             // (developer ablation, WRONG results: no partial sums written)
#endif
                if (m < Mo && n < a.N) out[(size_t)m * a.N + n] = acc[i][j][e];
            }
        }
}
// (what the plan asks before it sizes the partial sums: dg_api.hip head_splits)
bool dg_head_wgrad_one_pass(int M, int N, int M2, int P) {
#ifdef DG_DEVTOOLS
    if (const char* e = getenv("DG_HEAD_WGRAD3")) if (e[0] == '0') return false;
#endif
    return M > 256 && M <= 384 && M2 > 0 && M2 <= 128 && (P & 7) == 0;
}

template <typename TA, typename TB, typename TA2 = TA>
static hipError_t launch_wgrad(const DgHeadWgradArgs& a, hipStream_t s) {
    if constexpr (std::is_same<TA, __bf16>::value && std::is_same<TB, float>::value && std::is_same<TA2, float>::value) {
        if (a.A2h && dg_head_wgrad_one_pass(a.M, a.N, a.M2, a.P) && (a.splits & 7) == 0) {
            const int smem = W3_NSTAGE * W3_STAGE;
            hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_head_wgrad3), smem);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_head_wgrad3, dim3(((a.N + 127) / 128) * a.splits), dim3(512), smem, s, a);
            return hipGetLastError();
        }
    }
    const int tiles = ((a.N + 127) / 128) * ((a.M + 127) / 128 + (a.M2 + 127) / 128);
    if ((a.P & 7) == 0 && (a.splits & 7) == 0) {
        const int smem = 4 * 128 * (32 * 2 + 16);
        hipLaunchKernelGGL((k_head_wgrad2<TA, TB, TA2>), dim3(tiles * a.splits), dim3(256), smem, s, a);
        return hipGetLastError();
    }
    // (odd position counts, or fewer steps than eight splits: the direct form, one product per launch)
    dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, a.splits);
    hipLaunchKernelGGL((k_head_wgrad<TA, TB>), grid, dim3(256), 0, s, a);
    if (a.M2 > 0) {
        DgHeadWgradArgs b = a;
        b.A = a.A2; b.dA = a.dA2; b.M = a.M2; b.keep = a.keep_2; b.part = a.part2; b.M2 = 0;
        hipLaunchKernelGGL((k_head_wgrad<TA2, TB>), dim3((a.N + 127) / 128, (b.M + 127) / 128, a.splits), dim3(256), 0, s, b);
    }
    return hipGetLastError();
}
// a_bf16 / b_bf16: element types of A and Bm; a second product (A2: fp32) rides in the same launch when a.M2 > 0
hipError_t dg_launch_head_wgrad(const DgHeadWgradArgs& a, bool a_bf16, bool b_bf16, hipStream_t s) {
    if (a.M2 > 0) {
        if (a_bf16 && !b_bf16) return launch_wgrad<__bf16, float, float>(a, s);
        return hipErrorInvalidValue;
    }
    if (a_bf16 && !b_bf16) return launch_wgrad<__bf16, float>(a, s);
    if (!a_bf16 && !b_bf16) return launch_wgrad<float, float>(a, s);
    if (!a_bf16 && b_bf16) return launch_wgrad<float, __bf16>(a, s);
    return launch_wgrad<__bf16, __bf16>(a, s);
}

// Up to six reductions in one launch: out[i] (and out2[i]) = scale * sum over splits of part[split][i], in a fixed order.
// blockIdx.y = job; a block owns 64 consecutive outputs, its four waves take the splits k = wave, wave + 4, ... with eight loads
// in flight each and meet in LDS (one thread per output walking all the splits was a chain of splits / 4 memory latencies: 33 us
// for 70 MB).
__global__ __launch_bounds__(256) void k_head_reduce(const DgHeadReduceArgs a) {
    __shared__ float red[4][64];
    const DgHeadReduceJob& J = a.jobs[blockIdx.y];
    const int o = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 64 + o, n = J.n;
    if (blockIdx.x * 64 >= n) return;
    const float* part = J.part + (i < n ? i : 0);
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    int k = w;
    for (; k + 28 < J.splits; k += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(k + 4 * u) * n];
    }
    for (; k + 12 < J.splits; k += 16) {                   // (the remainder four at a time: one split per round trip was 4 round trips behind
        float v[4];                                        //  the 80 splits of k_head_wgrad3 - same sums in the same order: acc[0] takes them one by one)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = part[(size_t)(k + 4 * u) * n];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[0] += v[u];
    }
    for (; k < J.splits; k += 4) acc[0] += part[(size_t)k * n];
    red[w][o] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (w == 0 && i < n) {
        const float v = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) * J.scale;
        J.out[i] = v;
        if (J.out2) J.out2[i] = v;
    }
}
hipError_t dg_launch_head_reduce(const DgHeadReduceArgs& a, hipStream_t s) {
    int nmax = 0;
    for (int j = 0; j < a.njobs; ++j) nmax = a.jobs[j].n > nmax ? a.jobs[j].n : nmax;
    hipLaunchKernelGGL(k_head_reduce, dim3((nmax + 63) / 64, a.njobs), dim3(256), 0, s, a);
    return hipGetLastError();
}

// bias gradients: out[row] (and out2[row]) = sum over images and positions of X[image][row][:]   (one block per row, fixed order)
template <typename T>
__global__ __launch_bounds__(256) void k_head_rowsum(const T* __restrict__ X, float* __restrict__ out, float* __restrict__ out2, int B, int R, int P,
                                                     int Bs, long long dX) {
    __shared__ float red[256];
    const int row = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const T* x = X + dg_img_off(b, (long long)R * P, Bs, dX) + (size_t)row * P;
        for (int p = threadIdx.x; p < P; p += 256) s += (float)x[p];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { out[row] = red[0]; if (out2) out2[row] = red[0]; }
}
hipError_t dg_launch_head_rowsum(const void* X, bool bf16, float* out, float* out2, int B, int R, int P, hipStream_t s, int Bs, long long dX) {
    if (bf16) hipLaunchKernelGGL(k_head_rowsum<__bf16>, dim3(R), dim3(256), 0, s, static_cast<const __bf16*>(X), out, out2, B, R, P, Bs, dX);
    else hipLaunchKernelGGL(k_head_rowsum<float>, dim3(R), dim3(256), 0, s, static_cast<const float*>(X), out, out2, B, R, P, Bs, dX);
    return hipGetLastError();
}

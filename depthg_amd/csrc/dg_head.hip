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

// eight fp32 weights W[row][k .. k+3], W[row][k' .. k'+3] (k, k' = the lane's two element blocks) times keep flags -> bf16 fragment
__device__ __forceinline__ bf16x8 weight_frag(const float* __restrict__ W, const int ld, const int row, const int nrows, const int kbase,
                                              const int K, const float* keep /* LDS [Kpad] or null */, const int lane) {
    f32x4 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = kbase + frag_k_order(lane, u);
        if (row < nrows && k + 3 < K) v[u] = *reinterpret_cast<const f32x4*>(W + (size_t)row * ld + k);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = (row < nrows && k + e < K) ? W[(size_t)row * ld + k + e] : 0.f;
        }
        if (keep) {
            const f32x4 kp = *reinterpret_cast<const f32x4*>(keep + k);
            v[u] = v[u] * kp;
        }
    }
    return pack8(v[0], v[1]);
}

// MB = 16-row blocks of hidden channels per wave (Cpad = 64 MB), NT = positions per block
template <int MB, int NT>
__global__ __launch_bounds__(256) void k_head_fwd(const DgHeadFwdArgs a) {
    constexpr int CP = 64 * MB, NB = NT / 16, FROW = NT * 2 + 32, HROW = CP * 2 + 32, DBMAX = 8;
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    char* const Ft = hsm;                                   // [CP][FROW]  bf16 f tile, channel-major; later Hm [channel][position]
    char* const Ht = hsm + CP * FROW;                       // [NT][HROW]  bf16 hidden tile, position-major
    float* const km1 = reinterpret_cast<float*>(Ht + NT * HROW);     // [CP] keep flags of cluster1's dropout (1 everywhere without one)
    float* const km2 = km1 + CP;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.y, p0 = blockIdx.x * NT;
    const int C = a.C, D = a.D, P = a.P;
    const bool nonlinear = a.w2a != nullptr;
    const float s1 = a.keep1 ? a.scale : 1.f, s2 = a.keep2 ? a.scale : 1.f, s3 = a.keep3 ? a.scale : 1.f;

    // ---- stage the feature tile: fp32 (C, NT) -> bf16 LDS image, drop3(f) written on the way
    for (int k = tid; k < CP; k += 256) {
        km1[k] = k < C ? (a.keep1 ? a.keep1[(size_t)b * C + k] : 1.f) : 0.f;
        km2[k] = k < C ? (a.keep2 ? a.keep2[(size_t)b * C + k] : 1.f) : 0.f;
    }
    {
        constexpr int Q = NT / 4;
        const bool vec = (P & 3) == 0;
        for (int idx = tid; idx < CP * Q; idx += 256) {
            const int k = idx / Q, q4 = idx - k * Q, p = p0 + 4 * q4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < C) {
                const float* src = a.feat + ((size_t)b * C + k) * P + p;
                if (vec && p + 3 < P) v = *reinterpret_cast<const f32x4*>(src);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (p + e < P) v[e] = src[e];
                }
                if (a.feats_out) {
                    const float f3 = a.keep3 ? a.keep3[(size_t)b * C + k] * s3 : 1.f;
                    float* dst = a.feats_out + ((size_t)b * C + k) * P + p;
                    const f32x4 o = v * f3;
                    if (vec && p + 3 < P) *reinterpret_cast<f32x4*>(dst) = o;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (p + e < P) dst[e] = o[e];
                    }
                }
            }
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4*>(Ft + k * FROW + q4 * 8) = o4;
        }
    }
    __syncthreads();

    // ---- hidden = relu(s2 * W2a[:, kept] f + b2a): wave `wid` owns hidden channels [16 MB wid, 16 MB (wid + 1))
    f32x4 acc1[MB][NB];
    const int mbase = wid * 16 * MB;
    if (nonlinear) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < CP / 32; ++ks) {
            bf16x8 af[MB], bfr[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) af[i] = weight_frag(a.w2a, C, mbase + 16 * i + c16, C, 32 * ks, C, km2, lane);
#pragma unroll
            for (int j = 0; j < NB; ++j) bfr[j] = tr_frag(Ft, FROW, 32 * ks, 16 * j, lane);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc1[i][j], 0, 0, 0);
        }
    }
    // ---- cluster1 part of the code: W1[:, kept] f.  Wave -> position block nb, code-channel blocks mb0, mb0 + MSTEP, ...
    constexpr int MSTEP = 4 / NB;
    const int nb = wid % NB, mb0 = wid / NB, DB = (D + 15) / 16;
    f32x4 acc2a[DBMAX / MSTEP], acc2b[DBMAX / MSTEP];
#pragma unroll
    for (int i = 0; i < DBMAX / MSTEP; ++i) { acc2a[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2b[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int ks = 0; ks < CP / 32; ++ks) {
        const bf16x8 bfr = tr_frag(Ft, FROW, 32 * ks, 16 * nb, lane);
#pragma unroll
        for (int i = 0; i < DBMAX / MSTEP; ++i) {
            const int mb = mb0 + i * MSTEP;
            if (mb < DB) {
                const bf16x8 af = weight_frag(a.w1, C, 16 * mb + c16, D, 32 * ks, C, km1, lane);
                acc2a[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc2a[i], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                       // every wave is done reading the f tile
    if (nonlinear) {
        // bias + ReLU in the accumulators; hidden tile -> Ht [position][channel] (next operand) and Hm [channel][position] (saved)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int m = mbase + 16 * i + 4 * g;
            f32x4 bias;
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = m + r < C ? a.b2a[m + r] : 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int p = 16 * j + c16;
                bf16x4 h4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = m + r < C ? fmaxf(fmaf(acc1[i][j][r], s2, bias[r]), 0.f) : 0.f;
                    h4[r] = (__bf16)h;
                    *reinterpret_cast<__bf16*>(Ft + (m + r) * FROW + p * 2) = h4[r];
                }
                *reinterpret_cast<bf16x4*>(Ht + p * HROW + m * 2) = h4;
            }
        }
        __syncthreads();
        // ---- cluster2's output convolution: W2b hidden
        for (int ks = 0; ks < CP / 32; ++ks) {
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(Ht + (16 * nb + c16) * HROW + (32 * ks + 8 * g) * 2);
#pragma unroll
            for (int i = 0; i < DBMAX / MSTEP; ++i) {
                const int mb = mb0 + i * MSTEP;
                if (mb < DB) {
                    // (natural k order on both sides: element e of group g is k = 32 ks + 8 g + e)
                    const int row = 16 * mb + c16, k = 32 * ks + 8 * g;
                    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
                    if (row < D && k + 7 < C) {
                        lo = *reinterpret_cast<const f32x4*>(a.w2b + (size_t)row * C + k);
                        hi = *reinterpret_cast<const f32x4*>(a.w2b + (size_t)row * C + k + 4);
                    } else if (row < D) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { lo[e] = k + e < C ? a.w2b[(size_t)row * C + k + e] : 0.f; hi[e] = k + 4 + e < C ? a.w2b[(size_t)row * C + k + 4 + e] : 0.f; }
                    }
                    acc2b[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack8(lo, hi), bfr, acc2b[i], 0, 0, 0);
                }
            }
        }
    }
    // ---- code = s1 * (W1 f) + (W2b hidden) + b1 + b2b
#pragma unroll
    for (int i = 0; i < DBMAX / MSTEP; ++i) {
        const int mb = mb0 + i * MSTEP;
        if (mb < DB) {
            const int p = p0 + 16 * nb + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * mb + 4 * g + r;
                if (d < D && p < P) {
                    float v = fmaf(acc2a[i][r], s1, a.b1[d]);
                    if (nonlinear) v += acc2b[i][r] + a.b2b[d];
                    a.code[((size_t)b * D + d) * P + p] = v;
                }
            }
        }
    }
    // ---- hidden tile -> HBM (B, C, P) bf16, rows copied from the channel-major LDS image
    if (nonlinear && a.hidden) {
        constexpr int Q8 = NT / 8;
        const bool vec = (P & 7) == 0;
        for (int idx = tid; idx < C * Q8; idx += 256) {
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
}

template <int MB, int NT>
static hipError_t launch_head_fwd(const DgHeadFwdArgs& a, hipStream_t s) {
    constexpr int CP = 64 * MB;
    const int smem = CP * (NT * 2 + 32) + NT * (CP * 2 + 32) + 2 * CP * 4;
    auto kern = k_head_fwd<MB, NT>;
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((a.P + NT - 1) / NT, a.B), dim3(256), smem, s, a);
    return hipGetLastError();
}

hipError_t dg_launch_head_fwd(const DgHeadFwdArgs& a, hipStream_t s) {
    if (a.C <= 64) return launch_head_fwd<1, 64>(a, s);
    if (a.C <= 128) return launch_head_fwd<2, 64>(a, s);
    if (a.C <= 192) return launch_head_fwd<3, 64>(a, s);
    if (a.C <= 384) return launch_head_fwd<6, 64>(a, s);
    if (a.C <= 768) return launch_head_fwd<12, 32>(a, s);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------- backward
// d hidden_pre = (hidden > 0) * (W2b^T d code), bf16 (B, C, P).  Block = (position tile of 64, image); the product is formed
// TRANSPOSED (rows = positions, columns = hidden channels): an accumulator then holds four consecutive positions of one channel,
// which is 8 contiguous bytes of `hidden` (the ReLU mask) and of the output.

template <int MB>
__global__ __launch_bounds__(256) void k_head_dh(const DgHeadDhArgs a) {
    constexpr int NT = 64, FROW = NT * 2 + 32, DPMAX = 128;
    __shared__ __attribute__((aligned(16))) char Dt[DPMAX * FROW];      // [d][position] bf16, zero padded to a multiple of 32 rows
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.y, p0 = blockIdx.x * NT;
    const int C = a.C, D = a.D, P = a.P, DP = (D + 31) / 32 * 32;
    for (int idx = tid; idx < DP * (NT / 4); idx += 256) {
        const int d = idx / (NT / 4), q4 = idx - d * (NT / 4), p = p0 + 4 * q4;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)((d < D && p + e < P) ? a.gcode[((size_t)b * D + d) * P + p + e] : 0.f);
        *reinterpret_cast<bf16x4*>(Dt + d * FROW + q4 * 8) = o;
    }
    __syncthreads();
    f32x4 acc[4][MB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nbase = wid * 16 * MB;
    for (int ks = 0; ks < DP / 32; ++ks) {
        bf16x8 af[4], bfr[MB];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = tr_frag(Dt, FROW, 32 * ks, 16 * i, lane);     // A[position][d]: the same transposing read
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = nbase + 16 * j + c16;                                            // B[d][channel m] = W2b[d][m]
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int d0 = 32 * ks + frag_k_order(lane, u);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = (m < C && d0 + e < D) ? a.w2b[(size_t)(d0 + e) * C + m] : 0.f;
            }
            bfr[j] = pack8(v[0], v[1]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    const bool vec = (P & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = nbase + 16 * j + c16, p = p0 + 16 * i + 4 * g;
            if (m >= C || p >= P) continue;
            const size_t off = ((size_t)b * C + m) * P + p;
            bf16x4 h4, o4;
            if (vec) h4 = *reinterpret_cast<const bf16x4*>(a.hidden + off);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) h4[r] = p + r < P ? a.hidden[off + r] : (__bf16)0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = (float)h4[r] > 0.f ? (__bf16)acc[i][j][r] : (__bf16)0.f;
            if (vec) *reinterpret_cast<bf16x4*>(a.dh + off) = o4;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (p + r < P) a.dh[off + r] = o4[r];
            }
        }
}

hipError_t dg_launch_head_dh(const DgHeadDhArgs& a, hipStream_t s) {
    dim3 grid((a.P + 63) / 64, a.B);
#define DG_DH(MB_) { hipLaunchKernelGGL(k_head_dh<MB_>, grid, dim3(256), 0, s, a); return hipGetLastError(); }
    if (a.C <= 64) DG_DH(1)
    if (a.C <= 128) DG_DH(2)
    if (a.C <= 192) DG_DH(3)
    if (a.C <= 384) DG_DH(6)
    if (a.C <= 768) DG_DH(12)
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
    for (int s = s0; s < s1 && live; ++s) {
        const int b = s / steps_img, p = (s - b * steps_img) * 32 + 8 * g;
        const TA* Ab = static_cast<const TA*>(a.A) + (size_t)b * a.M * a.P;
        const TB* Bb = static_cast<const TB*>(a.Bm) + (size_t)b * a.N * a.P;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = row_frag<TA>(Ab, m0 + 16 * i + c16, a.M, a.P, p, vec);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 16 * j + c16;
            bfr[j] = row_frag<TB>(Bb, n, a.N, a.P, p, vec);
            if (a.keep && n < a.N && a.keep[(size_t)b * a.N + n] == 0.f) bfr[j] = bf16x8{};
        }
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

template <typename TA, typename TB>
static hipError_t launch_wgrad(const DgHeadWgradArgs& a, hipStream_t s) {
    dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, a.splits);
    hipLaunchKernelGGL((k_head_wgrad<TA, TB>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t dg_launch_head_wgrad(const DgHeadWgradArgs& a, bool a_bf16, bool b_bf16, hipStream_t s) {
    if (a_bf16 && !b_bf16) return launch_wgrad<__bf16, float>(a, s);
    if (!a_bf16 && !b_bf16) return launch_wgrad<float, float>(a, s);
    if (!a_bf16 && b_bf16) return launch_wgrad<float, __bf16>(a, s);
    return launch_wgrad<__bf16, __bf16>(a, s);
}

// out[i] = scale * sum over splits of part[split][i]   (fixed order)
__global__ void k_head_reduce(const float* __restrict__ part, float* __restrict__ out, int n, int splits, float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
    out[i] = s * scale;
}
hipError_t dg_launch_head_reduce(const float* part, float* out, int n, int splits, float scale, hipStream_t s) {
    hipLaunchKernelGGL(k_head_reduce, dim3((n + 255) / 256), dim3(256), 0, s, part, out, n, splits, scale);
    return hipGetLastError();
}

// bias gradients: out[row] (and out2[row]) = sum over images and positions of X[image][row][:]   (one block per row, fixed order)
template <typename T>
__global__ __launch_bounds__(256) void k_head_rowsum(const T* __restrict__ X, float* __restrict__ out, float* __restrict__ out2, int B, int R, int P) {
    __shared__ float red[256];
    const int row = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const T* x = X + ((size_t)b * R + row) * P;
        for (int p = threadIdx.x; p < P; p += 256) s += (float)x[p];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { out[row] = red[0]; if (out2) out2[row] = red[0]; }
}
hipError_t dg_launch_head_rowsum(const void* X, bool bf16, float* out, float* out2, int B, int R, int P, hipStream_t s) {
    if (bf16) hipLaunchKernelGGL(k_head_rowsum<__bf16>, dim3(R), dim3(256), 0, s, static_cast<const __bf16*>(X), out, out2, B, R, P);
    else hipLaunchKernelGGL(k_head_rowsum<float>, dim3(R), dim3(256), 0, s, static_cast<const float*>(X), out, out2, B, R, P);
    return hipGetLastError();
}

// Fused correlation-loss kernel for gfx950 (MI355X).
//
// One workgroup = NWAVES waves; each wave keeps 32 positions of the stationary operand "R"
// (normalised feats + code rows, bf16) in registers and walks over the streamed operand "S" in
// tiles of 32 positions staged through LDS.  Per 32x32 tile and wave:
//     Yf[s][r] = sum_k Sf[s][k] Rf[r][k]     (KF/16 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)
//     Yc[s][r] = sum_d Sc[s][d] Rc[r][d]     (KD/16 MFMAs)
//     epilogue (registers only): centering, shift, clamp, loss / cd partial sums, G = dLoss/dcd
//     dR[r][:] += sum_s G[s][r] ScP[s][:]    (accumulator tile reused as the A operand, 2*DP/32 MFMAs)
// The (B,P,P) tensors fd / cd / loss of the reference (src/modules.py:1231-1254) are never
// written to HBM unless a caller asks for them (materialise path).
//
// Reference semantics reproduced here: helper() src/modules.py:1231-1254,
// depth_feature_correlation() :1256-1278 (job kind DG_JOB_DEPTH), norm() :789-790 (backward part).
#include "dg_common.h"

__device__ __forceinline__ bf16x8 lds_read_frag(const char* base, int byte_off) {
    return *reinterpret_cast<const bf16x8*>(base + byte_off);
}
__device__ __forceinline__ f16x8 lds_read_frag_h(const char* base, int byte_off) {
    return *reinterpret_cast<const f16x8*>(base + byte_off);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the 32 lanes of each half-wave (lanes 0-31 and 32-63 separately)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LDS image of one S tile (32 positions).  Row strides are an odd number of 16-byte granules so
// that the 16 lanes of a ds_read_b128 group (distinct rows, same granule) hit distinct bank slots.
template <int NKF, int NKD>
struct TileLayout {
    static constexpr int KF = NKF * 16, KD = NKD * 16;
    static constexpr int GF = KF / 8, GD = KD / 8;            // granules per row
    static constexpr int SF_STRIDE = (GF | 1) * 16;           // bytes
    static constexpr int SC_STRIDE = (GD | 1) * 16;
    static constexpr int SP_STRIDE = 5 * 16;                  // 32 positions * 2 B + 16 pad
    static constexpr int OFF_SF = 0;
    static constexpr int OFF_SC = OFF_SF + 32 * SF_STRIDE;
    static constexpr int OFF_SP = OFF_SC + 32 * SC_STRIDE;
    static constexpr int OFF_RV = OFF_SP + KD * SP_STRIDE;    // 32 floats rvec of the tile rows
    static constexpr int OFF_NZ = OFF_RV + 128;               // 32 floats depth indicator of the tile rows
    static constexpr int BYTES = OFF_NZ + 128;
};

template <int NKF, int NKD, int NWAVES, bool GRAD>
__global__ __launch_bounds__(NWAVES * 64) void k_corr_main(const DgCorrArgs args) {
    using L = TileLayout<NKF, NKD>;
    constexpr int KF = L::KF, KD = L::KD;
    constexpr int NDF = KD / 32;            // 32-wide output fragments of dR
    constexpr int DP = KD;                  // padded code width of dR
    constexpr int NT = NWAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tile = smem;
    float* red = reinterpret_cast<float*>(smem + L::BYTES);  // [NWAVES][2]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // ---- block -> (job, image, row block)
    int bid = blockIdx.x;
    const int per_job = args.B * args.nrb;
    const int jid = bid / per_job;
    bid -= jid * per_job;
    const int n = bid / args.nrb;
    const int rb = bid - n * args.nrb;
    const DgJob& job = args.jobs[jid];
    const int Ppad = args.Ppad, P = args.P;
    const bool depth_job = job.kind == DG_JOB_DEPTH;
    const int nR = job.ridx ? (int)job.ridx[n] : n;
    const int nS = job.sidx ? (int)job.sidx[n] : n;

    const int row0 = rb * (NWAVES * 32) + wid * 32;
    const bool wave_active = row0 < Ppad;
    const int pr = wave_active ? row0 + r : 0;   // stationary position of this lane (clamped for idle waves)

    // ---- stationary operand fragments -> registers
    bf16x8 Rf[NKF];
    f16x8 Rc[NKD];
    if (!depth_job) {
        const uint16_t* src = job.Rf + ((size_t)nR * Ppad + pr) * KF + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NKF; ++ks) Rf[ks] = *reinterpret_cast<const bf16x8*>(src + 16 * ks);
    } else {
#pragma unroll
        for (int ks = 0; ks < NKF; ++ks) Rf[ks] = bf16x8{};
    }
    {
        const uint16_t* src = job.Rc + ((size_t)nR * Ppad + pr) * KD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) Rc[ks] = *reinterpret_cast<const f16x8*>(src + 16 * ks);
    }

    // ---- per-job scalars
    float c0 = -job.shift;    // fd'' - shift = Yf - rowmean + (m0 - shift)
    if (job.rvec) {
        float ms = 0.f;
        for (int b = 0; b < args.B; ++b) ms += job.rsum[b];
        c0 += ms * args.inv_BP;
    }
    const bool on_lane = job.center_on_lane != 0;
    float cen_lane = 0.f, nz_lane = 0.f;
    if (on_lane && job.rvec) cen_lane = job.rvec[(size_t)n * Ppad + pr];
    if (depth_job) nz_lane = job.nzR[(size_t)n * Ppad + pr];
    const float lo = args.lo, hi = args.hi;

    f32x16 dR[NDF];
#pragma unroll
    for (int f = 0; f < NDF; ++f) dR[f] = f32x16{};
    float lsum = 0.f, csum = 0.f;

    const int ntiles = Ppad / 32;
    for (int t = 0; t < ntiles; ++t) {
        const int s0 = t * 32;
        __syncthreads();   // previous tile fully consumed
        // ---- stage S tile: global -> LDS (16-byte granules, coalesced along K)
        if (!depth_job) {
            const uint16_t* g = job.Sf + ((size_t)nS * Ppad + s0) * KF;
            for (int id = tid; id < 32 * L::GF; id += NT) {
                int q = id / L::GF, gg = id - q * L::GF;
                uint4 v = *reinterpret_cast<const uint4*>(g + (size_t)q * KF + gg * 8);
                *reinterpret_cast<uint4*>(tile + L::OFF_SF + q * L::SF_STRIDE + gg * 16) = v;
            }
        }
        {
            const uint16_t* g = job.Sc + ((size_t)nS * Ppad + s0) * KD;
            for (int id = tid; id < 32 * L::GD; id += NT) {
                int q = id / L::GD, gg = id - q * L::GD;
                uint4 v = *reinterpret_cast<const uint4*>(g + (size_t)q * KD + gg * 8);
                *reinterpret_cast<uint4*>(tile + L::OFF_SC + q * L::SC_STRIDE + gg * 16) = v;
            }
        }
        if (GRAD) {
            const uint16_t* g = job.ScP + (size_t)nS * KD * Ppad + s0;
            for (int id = tid; id < KD * 4; id += NT) {
                int d = id >> 2, gg = id & 3;
                uint4 v = *reinterpret_cast<const uint4*>(g + (size_t)d * Ppad + gg * 8);
                *reinterpret_cast<uint4*>(tile + L::OFF_SP + d * L::SP_STRIDE + gg * 16) = v;
            }
        }
        if (tid < 32) {
            float rv = 0.f, nz = 0.f;
            if (!on_lane && job.rvec) rv = job.rvec[(size_t)n * Ppad + s0 + tid];
            if (depth_job) nz = job.nzS[(size_t)n * Ppad + s0 + tid];
            reinterpret_cast<float*>(tile + L::OFF_RV)[tid] = rv;
            reinterpret_cast<float*>(tile + L::OFF_NZ)[tid] = nz;
        }
        __syncthreads();

        if (wave_active) {
            // ---- correlations on the matrix cores
            f32x16 Yf = f32x16{}, Yc = f32x16{};
            if (!depth_job) {
                const char* base = tile + L::OFF_SF + r * L::SF_STRIDE + h * 16;
#pragma unroll
                for (int ks = 0; ks < NKF; ++ks)
                    Yf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read_frag(base, ks * 32), Rf[ks], Yf, 0, 0, 0);
            }
            {
                const char* base = tile + L::OFF_SC + r * L::SC_STRIDE + h * 16;
#pragma unroll
                for (int ks = 0; ks < NKD; ++ks)
                    Yc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_read_frag_h(base, ks * 32), Rc[ks], Yc, 0, 0, 0);
            }
            // ---- epilogue: element i of the accumulator is (tile row s = (i&3)+8*(i>>2)+4*h, column r)
            const float* rvs = reinterpret_cast<const float*>(tile + L::OFF_RV);
            const float* nzs = reinterpret_cast<const float*>(tile + L::OFF_NZ);
            float g[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int sl = (i & 3) + 8 * (i >> 2) + 4 * h;
                float fdv;
                if (depth_job) fdv = nz_lane * nzs[sl] + c0;
                else           fdv = Yf[i] - (on_lane ? cen_lane : rvs[sl]) + c0;
                const float cdv = Yc[i];
                const float cl = fminf(fmaxf(cdv, lo), hi);
                lsum = fmaf(cl, fdv, lsum);
                csum += cdv;
                g[i] = (cdv >= lo && cdv <= hi) ? -fdv : 0.f;
                if (job.out_cd || job.out_loss) {   // materialise (R = operand 2 on lanes -> coalesced along q)
                    const int p = s0 + sl, q = pr;
                    if (p < P && q < P) {
                        size_t o = ((size_t)n * P + p) * P + q;
                        if (job.out_cd) job.out_cd[o] = depth_job ? nz_lane * nzs[sl] : cdv;
                        if (job.out_loss) job.out_loss[o] = -cl * fdv;
                    }
                }
            }
            if (GRAD) {
                // ---- dR[r][:] += sum_s G[s][r] * ScP[s][:]   (accumulator tile as A operand)
                f16x8 ga[2];
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int j = 0; j < 8; ++j) ga[sp][j] = (_Float16)g[8 * sp + j];
#pragma unroll
                for (int f = 0; f < NDF; ++f) {
                    const char* base = tile + L::OFF_SP + (32 * f + r) * L::SP_STRIDE + h * 16;
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp)
                        dR[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[sp], lds_read_frag_h(base, sp * 32), dR[f], 0, 0, 0);
                }
            }
        }
    }

    // ---- partial sums of this block (deterministic two-level reduction; finished by k_corr_finish)
    lsum = wave_sum(lsum);
    csum = wave_sum(csum);
    __syncthreads();
    if (lane == 0) { red[wid * 2] = wave_active ? lsum : 0.f; red[wid * 2 + 1] = wave_active ? csum : 0.f; }
    __syncthreads();
    if (tid == 0 && job.part) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < NWAVES; ++w) { a += red[w * 2]; b += red[w * 2 + 1]; }
        job.part[(size_t)(n * args.nrb + rb) * 2] = a;
        job.part[(size_t)(n * args.nrb + rb) * 2 + 1] = b;
    }

    // ---- normalisation backward and store:  dc = (dx - x <x,dx>) / max(||c||, eps)
    if (GRAD && job.dR && wave_active) {
        // dR[f][i] is (stationary row rr = row0 + (i&3)+8*(i>>2)+4*h, code channel d = 32 f + r)
        float xv[NDF][16];
        float dot[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) dot[i] = 0.f;
#pragma unroll
        for (int f = 0; f < NDF; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float x = (float)reinterpret_cast<const _Float16*>(job.Rc)[((size_t)nR * Ppad + rr) * KD + 32 * f + r];
                xv[f][i] = x;
                dot[i] = fmaf(x, dR[f][i], dot[i]);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) dot[i] = half_sum(dot[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (rr < P) {
                const float inv = job.RcInv[(size_t)nR * Ppad + rr];
#pragma unroll
                for (int f = 0; f < NDF; ++f) {
                    const int d = 32 * f + r;
                    job.dR[((size_t)n * Ppad + rr) * DP + d] = (dR[f][i] - xv[f][i] * dot[i]) * inv;
                }
            }
        }
    }
}

// Final reduction of the per-block partial sums into the 8 output scalars.

__global__ void k_corr_finish(const DgFinishArgs a) {
    __shared__ double acc[8];
    __shared__ double wred[8][2];
    const int tid = threadIdx.x;
    if (tid < 8) acc[tid] = 0.0;
    __syncthreads();
    for (int j = 0; j < a.njobs; ++j) {
        double l = 0.0, c = 0.0;
        for (int i = tid; i < a.nblk[j]; i += blockDim.x) { l += a.part[j][2 * i]; c += a.part[j][2 * i + 1]; }
        for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o, 64); c += __shfl_xor(c, o, 64); }
        if ((tid & 63) == 0) { wred[tid >> 6][0] = l; wred[tid >> 6][1] = c; }
        __syncthreads();
        if (tid == 0) {
            double ls = 0, cs = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ls += wred[w][0]; cs += wred[w][1]; }
            if (a.slot_loss[j] >= 0) acc[a.slot_loss[j]] += -ls * (double)a.scale[j];
            if (a.slot_cd[j] >= 0) acc[a.slot_cd[j]] += cs * (double)a.scale[j];
        }
        __syncthreads();
    }
    if (a.nz) {   // mean(dd) = mean_n (sum_p nz[n][p])^2 / P^2
        double m = 0.0;
        for (int n = 0; n < a.B; ++n) {
            double s = 0.0;
            for (int p = tid; p < a.P; p += blockDim.x) s += a.nz[(size_t)n * a.Ppad + p];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if ((tid & 63) == 0) wred[tid >> 6][0] = s;
            __syncthreads();
            if (tid == 0) { double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wred[w][0]; m += t * t; }
            __syncthreads();
        }
        if (tid == 0) acc[7] = m / ((double)a.B * a.P * a.P);
    }
    __syncthreads();
    if (tid < 8) a.out[tid] = (float)acc[tid];
}

// ---- launch helpers (host) ------------------------------------------------------------------
template <int NKF, int NKD, int NWAVES, bool GRAD>
static hipError_t launch_corr_t(const DgCorrArgs& args, hipStream_t stream) {
    using L = TileLayout<NKF, NKD>;
    const int smem = L::BYTES + NWAVES * 2 * 4;
    auto kern = k_corr_main<NKF, NKD, NWAVES, GRAD>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    const int grid = args.njobs * args.B * args.nrb;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), smem, stream, args);
    return hipGetLastError();
}

// KF in {128, 384, 768}, KD in {96, 128}; waves per block chosen by the caller (4 or 8).
hipError_t dg_launch_corr(const DgCorrArgs& args, int KF, int KD, int nwaves, bool grad, hipStream_t stream) {
#define DG_CASE(NKF_, NKD_, NW_)                                                                     \
    if (KF == NKF_ * 16 && KD == NKD_ * 16 && nwaves == NW_)                                         \
        return grad ? launch_corr_t<NKF_, NKD_, NW_, true>(args, stream) : launch_corr_t<NKF_, NKD_, NW_, false>(args, stream);
    DG_CASE(8, 6, 4) DG_CASE(8, 6, 8) DG_CASE(8, 8, 4) DG_CASE(8, 8, 8)
    DG_CASE(24, 6, 4) DG_CASE(24, 6, 8) DG_CASE(24, 8, 4) DG_CASE(24, 8, 8)
    DG_CASE(48, 6, 4) DG_CASE(48, 8, 4)
#undef DG_CASE
    return hipErrorInvalidValue;
}

hipError_t dg_launch_finish(const DgFinishArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_corr_finish, dim3(1), dim3(256), 0, stream, a);
    return hipGetLastError();
}

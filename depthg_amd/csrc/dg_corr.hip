// Fused correlation-loss kernel for gfx950 (MI355X).
//
// One workgroup = NWAVES waves; each wave keeps 32 positions of the stationary operand "R"
// (normalised feats bf16 + code fp16 rows) in registers and walks over the streamed operand "S" in
// tiles of 32 positions.  A tile is one contiguous blob in HBM (dg_common.h) that is DMA'd into LDS
// (global_load_lds_dwordx4) one tile ahead of the computation; one workgroup barrier per tile.
// Per 32x32 tile and wave:
//     Yf[s][r] = sum_k Sf[s][k] Rf[r][k]     (KF/16 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)
//     Yc[s][r] = sum_d Sc[s][d] Rc[r][d]     (KD/16 x v_mfma_f32_32x32x16_f16)
//     epilogue (registers only): centering, shift, clamp, loss / cd partial sums, G = dLoss/dcd
//     dR[r][:] += sum_s G[s][r] ScP[s][:]    (accumulator tile reused as the A operand, 2*KD/32 MFMAs)
// The (B,P,P) tensors fd / cd / loss of the reference (src/modules.py:1231-1254) are never
// written to HBM unless a caller asks for them (materialise path).
//
// Reference semantics reproduced here: helper() src/modules.py:1231-1254,
// depth_feature_correlation() :1256-1278 (job kind DG_JOB_DEPTH), norm() :789-790 (backward part).
#include "dg_common.h"
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lptr_t;

// LDS byte address (wave-uniform) of a pointer into the dynamic shared segment
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)p);
}

// LDS-DMA: every lane gives its own global source address; the wave writes 64 x 16 (or 64 x 4) contiguous
// bytes at the wave-uniform LDS address.  Issued through inline asm so that hipcc neither drains it with
// vmcnt(0) before unrelated LDS reads nor counts it; completion is enforced by the explicit
// "s_waitcnt vmcnt(0)" + s_barrier at the top of the tile loop (cdna guide 5.7: M0 written in the same statement).
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
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

enum { KIND_LANE = 0, KIND_ROW = 1, KIND_DEPTH = 2 };   // centering vector on lanes (R = operand 1) / on tile rows / depth term

// epilogue of one 32x32 tile: element i of the accumulators is (tile row s = (i&3)+8*(i>>2)+4*h, column r)
template <int KIND, bool SIMPLE, bool MAT>
__device__ __forceinline__ void tile_epilogue(const f32x16& Yf, const f32x16& Yc, const float* rvs, const float* nzs, int h,
                                              float c0, float c0_lane, float nz_lane, float lo, float hi,
                                              float& lsum, float& csum, float (&g)[16],
                                              const DgJob& job, size_t out_base, int p0, int P, bool q_ok, bool has_vec) {
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
        // rows 8*i4 + 4*h + (0..3): the per-row vector comes as one 16-byte LDS read
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND == KIND_ROW && has_vec) v4 = *reinterpret_cast<const float4*>(rvs + 8 * i4 + 4 * h);
        if (KIND == KIND_DEPTH) v4 = *reinterpret_cast<const float4*>(nzs + 8 * i4 + 4 * h);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * i4 + e;
            float fdv;
            if (KIND == KIND_DEPTH)     fdv = fmaf(nz_lane, vv[e], c0);
            else if (KIND == KIND_ROW)  fdv = Yf[i] + (c0 - vv[e]);
            else                        fdv = Yf[i] + c0_lane;
            const float cdv = Yc[i];
            csum += cdv;
            float gi, li;
            if (SIMPLE) {                    // zero_clamp, no stabalize: clamp(cd) = cd * mask
                gi = cdv >= 0.f ? -fdv : 0.f;
                li = gi * cdv;               // = -clamp(cd) * (fd - shift)
                lsum -= li;
            } else {
                const float cl = fminf(fmaxf(cdv, lo), hi);
                lsum = fmaf(cl, fdv, lsum);
                gi = (cdv >= lo && cdv <= hi) ? -fdv : 0.f;
                li = -cl * fdv;
            }
            g[i] = gi;
            if (MAT) {   // R = operand 2 on lanes -> stores contiguous along q
                const int p = p0 + 8 * i4 + 4 * h + e;
                if (q_ok && p < P) {
                    const size_t o = out_base + (size_t)p * P;
                    if (job.out_cd) job.out_cd[o] = KIND == KIND_DEPTH ? nz_lane * vv[e] : cdv;
                    if (job.out_loss) job.out_loss[o] = li;
                }
            }
        }
    }
}

template <int NKF, int NKD, int NWAVES, bool GRAD, bool MAT, bool SIMPLE, int KIND>
__device__ __forceinline__ void corr_body(const DgCorrArgs& args, const DgJob& job, const int n, const int rb, char* smem) {
    using BL = BlobT<NKF, NKD>;
    constexpr int KD = BL::KD, GF = BL::GF;
    constexpr int NDF = KD / 32;            // 32-wide output fragments of dR
    constexpr int DP = KD;                  // padded code width of dR
    constexpr int BUF = BL::BYTES + 256;    // blob + 32 rvec floats + 32 nz floats
    constexpr int RCB = BL::OFF_P - BL::OFF_C;   // bytes of one C part (the stationary code rows of one wave)
    constexpr int NBUF = BL::BYTES > 48 * 1024 ? 2 : 3;   // LDS buffers; tiles are fetched NBUF-1 ahead

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int Ppad = args.Ppad, P = args.P;
    const int ntiles = Ppad >> 5;
    const int nR = job.ridx ? (int)job.ridx[n] : n;
    const int nS = job.sidx ? (int)job.sidx[n] : n;

    const int rtile = rb * NWAVES + wid;             // 32-row tile of R owned by this wave
    const bool wave_active = rtile < ntiles;
    const int row0 = rtile * 32;
    const int pr = wave_active ? row0 + r : 0;       // stationary position of this lane (clamped for idle waves)

    // ---- stationary operand: feats fragments -> registers, code rows -> LDS (DMA of the C part of its blob)
    const char* Rblob = job.Rop + ((size_t)nR * ntiles + (wave_active ? rtile : 0)) * BL::BYTES;
    constexpr bool RCREG = NWAVES == 4;   // 4-wave blocks run one wave per SIMD (512 registers): keep the code rows there
    char* rc_lds = smem + NBUF * BUF + wid * RCB;
    const uint32_t smem_a = lds_addr(smem);
    f16x8 Rc[RCREG ? NKD : 1];
    if (RCREG) {
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) {
            Rc[ks] = *reinterpret_cast<const f16x8*>(Rblob + BL::OFF_C + ((2 * ks + h) * 32 + r) * 16);
            asm volatile("" : "+v"(Rc[ks]));
        }
    } else {
        for (int c = 0; c < RCB / 1024; ++c)
            dma16(Rblob + BL::OFF_C + c * 1024 + lane * 16, smem_a + NBUF * BUF + wid * RCB + c * 1024);
    }
    bf16x8 Rf[NKF];
    if (KIND != KIND_DEPTH) {
#pragma unroll
        for (int ks = 0; ks < NKF; ++ks)
            Rf[ks] = *reinterpret_cast<const bf16x8*>(Rblob + (r * GF + ((2 * ks + h) ^ (r & 15))) * 16);
        // make hipcc wait for these loads HERE: its counted vmcnt waits at their first use inside the tile loop
        // would also count (and drain) the tile DMAs it does not know about
#pragma unroll
        for (int ks = 0; ks < NKF; ++ks) asm volatile("" : "+v"(Rf[ks]));
    }

    // ---- per-job scalars
    float c0 = -job.shift;    // fd'' - shift = Yf - rowmean + (m0 - shift)
    if (KIND != KIND_DEPTH && job.rvec) c0 += job.m0[0];
    float c0_lane = c0, nz_lane = 0.f;
    if (KIND == KIND_LANE && job.rvec) c0_lane -= job.rvec[(size_t)n * Ppad + pr];
    if (KIND == KIND_DEPTH) nz_lane = job.nzR[(size_t)n * Ppad + pr];
    const float lo = args.lo, hi = args.hi;
    const bool has_vec = job.rvec != nullptr;     // no pointwise centering -> the row vector is all zeros

    // ---- tile staging by LDS-DMA (1 KiB per wave instruction, linear in HBM and in LDS)
    const char* Sbase = job.Sop + (size_t)nS * ntiles * BL::BYTES + lane * 16;
    const float* vsrc = reinterpret_cast<const float*>(args.dummy);      // per-row vector of the tile (rvec or nz)
    if (KIND == KIND_ROW && job.rvec) vsrc = job.rvec + (size_t)n * Ppad;
    if (KIND == KIND_DEPTH) vsrc = job.nzS + (size_t)n * Ppad;
    constexpr int C_BEGIN = KIND == KIND_DEPTH ? BL::CHUNK_C0 : 0;
    constexpr int C_END = GRAD ? BL::CHUNKS : BL::CHUNK_P0;
    auto issue = [&](int t, int b) {
        const char* src = Sbase + (size_t)t * BL::BYTES;
        const uint32_t dst = smem_a + b * BUF;
        for (int c = C_BEGIN + wid; c < C_END; c += NWAVES) dma16(src + c * 1024, dst + c * 1024);
        if (KIND != KIND_LANE && wid == NWAVES - 1)     // 32 floats of the tile rows (lanes 32-63 write a copy behind them)
            dma4(vsrc + t * 32 + (lane & 31), dst + BL::BYTES);
    };
    // DMA instructions this wave issues per tile (wave-uniform): the counted wait leaves exactly one tile in flight
    const int my_dma = (C_END - C_BEGIN - wid + NWAVES - 1) / NWAVES + ((KIND != KIND_LANE && wid == NWAVES - 1) ? 1 : 0);

    f32x16 dR[NDF];
#pragma unroll
    for (int f = 0; f < NDF; ++f) dR[f] = f32x16{};
    float lsum = 0.f, csum = 0.f;
    const size_t out_base = (size_t)n * P * P + pr;
    const bool q_ok = pr < P;

    int swz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) swz[j] = ((2 * j + h) ^ (r & 15)) << 4;

    // NBUF LDS buffers, tiles are fetched NBUF-1 ahead: with 3 buffers the DMAs of tile t+1 stay in flight at the top
    // of iteration t (counted vmcnt), with 2 buffers everything outstanding is tile t itself
    issue(0, 0);
    if (NBUF == 3 && ntiles > 1) issue(1, 1);
    int bcur = 0;
    for (int t = 0; t < ntiles; ++t) {
        // tile t has landed (this wave's pieces: vmcnt; everybody's: barrier); the buffer of tile t-1 is free again
        if (NBUF == 3 && t + 1 < ntiles) {
            switch (my_dma) {
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
                case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
                case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
                case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
                case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
                case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + NBUF - 1 < ntiles && !(args.debug & 1)) issue(t + NBUF - 1, bcur >= 1 ? bcur - 1 : NBUF - 1);   // the buffer tile t-1 used
        const char* tile = smem + bcur * BUF;
        bcur = bcur == NBUF - 1 ? 0 : bcur + 1;

        if (wave_active) {
            // ---- correlations on the matrix cores
            f32x16 Yf = f32x16{}, Yc = f32x16{};
            if (KIND != KIND_DEPTH && !(args.debug & 2)) {
                // granule 2ks+h of row r sits at slot (2ks+h) ^ (r&15): the XOR only touches the low 4 bits, so the
                // 8 k-steps of a 16-granule group share 8 per-lane offsets and the group index is an immediate
                const char* base = tile + r * (GF * 16);
#pragma unroll
                for (int ks = 0; ks < NKF; ++ks) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(base + swz[ks & 7] + (ks >> 3) * 256);
                    Yf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Rf[ks], Yf, 0, 0, 0);
                }
            }
            {
                const char* base = tile + BL::OFF_C + (h * 32 + r) * 16;
                const char* rbase = rc_lds + (h * 32 + r) * 16;
#pragma unroll
                for (int ks = 0; ks < NKD; ++ks) {
                    const f16x8 a = *reinterpret_cast<const f16x8*>(base + ks * 1024);
                    const f16x8 b = RCREG ? Rc[ks] : *reinterpret_cast<const f16x8*>(rbase + ks * 1024);
                    Yc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, Yc, 0, 0, 0);
                }
            }
            const float* rvs = reinterpret_cast<const float*>(tile + BL::BYTES);
            float g[16];
            tile_epilogue<KIND, SIMPLE, MAT>(Yf, Yc, rvs, rvs, h, c0, c0_lane, nz_lane, lo, hi, lsum, csum, g, job, out_base, t * 32, P, q_ok, has_vec);
            if (GRAD && !(args.debug & 4)) {
                // ---- dR[r][:] += sum_s G[s][r] * ScP[s][:]   (accumulator tile as A operand)
                f16x8 ga[2];
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int j = 0; j < 8; ++j) ga[sp][j] = (_Float16)g[8 * sp + j];
#pragma unroll
                for (int f = 0; f < NDF; ++f) {
                    const char* base = tile + BL::OFF_P + (h * KD + 32 * f + r) * 16;
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp) {
                        const f16x8 b = *reinterpret_cast<const f16x8*>(base + sp * (2 * KD * 16));
                        dR[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[sp], b, dR[f], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- partial sums of this block (deterministic two-level reduction; finished by k_corr_finish)
    float* red = reinterpret_cast<float*>(smem + NBUF * BUF + (RCREG ? 0 : NWAVES * RCB));
    lsum = wave_sum(lsum);
    csum = wave_sum(csum);
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
        float dot[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) dot[i] = 0.f;
#pragma unroll
        for (int f = 0; f < NDF; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = (i & 3) + 8 * (i >> 2) + 4 * h, d = 32 * f + r;
                const char* xb = RCREG ? Rblob + BL::OFF_C : rc_lds;
                const float x = (float)*reinterpret_cast<const _Float16*>(xb + ((d >> 3) * 32 + q) * 16 + (d & 7) * 2);
                dot[i] = fmaf(x, dR[f][i], dot[i]);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) dot[i] = half_sum(dot[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = (i & 3) + 8 * (i >> 2) + 4 * h, rr = row0 + q;
            if (rr < P) {
                const float inv = job.RcInv[(size_t)nR * Ppad + rr];
#pragma unroll
                for (int f = 0; f < NDF; ++f) {
                    const int d = 32 * f + r;
                    const char* xb = RCREG ? Rblob + BL::OFF_C : rc_lds;
                    const float x = (float)*reinterpret_cast<const _Float16*>(xb + ((d >> 3) * 32 + q) * 16 + (d & 7) * 2);
                    job.dR[((size_t)n * Ppad + rr) * DP + d] = (dR[f][i] - x * dot[i]) * inv;
                }
            }
        }
    }
}

template <int NKF, int NKD, int NWAVES, bool GRAD, bool MAT, bool SIMPLE>
__global__ __launch_bounds__(NWAVES * 64) void k_corr_main(const DgCorrArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUF] + Rc[NWAVES][C part] + red[NWAVES][2]
    // ---- XCD-aware block order: blocks that share an XCD (orig % 8) get a contiguous range of logical ids,
    //      so the row blocks of one (pair-set, image) stream the same S blobs through one L2.
    int bid;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    }
    const int per_img = args.njobs * args.nrb;      // image-major: the jobs of one image stream the same operand blobs
    const int n = bid / per_img;
    bid -= n * per_img;
    const int jid = bid / args.nrb;
    const int rb = bid - jid * args.nrb;
    const DgJob& job = args.jobs[jid];
    if (job.kind == DG_JOB_DEPTH) corr_body<NKF, NKD, NWAVES, GRAD, MAT, SIMPLE, KIND_DEPTH>(args, job, n, rb, smem);
    else if (job.center_on_lane == 0) corr_body<NKF, NKD, NWAVES, GRAD, MAT, SIMPLE, KIND_ROW>(args, job, n, rb, smem);
    else if (!MAT) corr_body<NKF, NKD, NWAVES, GRAD, MAT, SIMPLE, KIND_LANE>(args, job, n, rb, smem);
}

// ---- final reduction of the per-block partial sums into the 8 output scalars (two tiny launches)
// stage 1: one block per job (+ one for mean(dd)); stage 2: one thread combines the job sums in a fixed order.
__global__ __launch_bounds__(256) void k_corr_finish1(const DgFinishArgs a) {
    __shared__ double wred[4][2];
    const int tid = threadIdx.x, j = blockIdx.x;
    double l = 0.0, c = 0.0;
    if (j < a.njobs) {
        for (int i = tid; i < a.nblk[j]; i += 256) { l += a.part[j][2 * i]; c += a.part[j][2 * i + 1]; }
    } else if (a.nz) {   // mean(dd) = mean_n (sum_p nz[n][p])^2 / P^2 ; one wave per image, summed below
        for (int n = tid >> 6; n < a.B; n += 4) {
            double s = 0.0;
            for (int p = tid & 63; p < a.P; p += 64) s += a.nz[(size_t)n * a.Ppad + p];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if ((tid & 63) == 0) l += s * s;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o, 64); c += __shfl_xor(c, o, 64); }
    if ((tid & 63) == 0) { wred[tid >> 6][0] = l; wred[tid >> 6][1] = c; }
    __syncthreads();
    if (tid == 0) {
        a.jobsum[2 * j] = wred[0][0] + wred[1][0] + wred[2][0] + wred[3][0];
        a.jobsum[2 * j + 1] = wred[0][1] + wred[1][1] + wred[2][1] + wred[3][1];
    }
}

__global__ void k_corr_finish2(const DgFinishArgs a) {
    if (threadIdx.x != 0) return;
    double acc[DG_OUT_COUNT];
    for (int i = 0; i < DG_OUT_COUNT; ++i) acc[i] = 0.0;
    for (int j = 0; j < a.njobs; ++j) {
        if (a.slot_loss[j] >= 0) acc[a.slot_loss[j]] += -a.jobsum[2 * j] * (double)a.scale[j];
        if (a.slot_cd[j] >= 0) acc[a.slot_cd[j]] += a.jobsum[2 * j + 1] * (double)a.scale[j];
    }
    if (a.nz) acc[DG_OUT_DD] = a.jobsum[2 * a.njobs] / ((double)a.B * a.P * a.P);
    for (int i = 0; i < DG_OUT_COUNT; ++i) a.out[i] = (float)acc[i];
}

// ---- launch helpers (host) ------------------------------------------------------------------
template <int NKF, int NKD, int NWAVES, bool GRAD, bool MAT, bool SIMPLE>
static hipError_t launch_corr_t(const DgCorrArgs& args, hipStream_t stream) {
    using BL = BlobT<NKF, NKD>;
    const int nbuf = BL::BYTES > 48 * 1024 ? 2 : 3;
    const int smem = nbuf * (BL::BYTES + 256) + (NWAVES == 4 ? 0 : NWAVES * (BL::OFF_P - BL::OFF_C)) + NWAVES * 2 * 4;
    auto kern = k_corr_main<NKF, NKD, NWAVES, GRAD, MAT, SIMPLE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    const int grid = args.njobs * args.B * args.nrb;
    DgCorrArgs a2 = args;
    if (const char* e = getenv("DG_DEBUG")) a2.debug = atoi(e);   // developer ablation switches (timing only, results invalid)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), smem, stream, a2);
    return hipGetLastError();
}

// KF in {128, 384, 768}, KD in {96, 128}; waves per block chosen by the caller (4 or 8).
// mode: 0 = loss only, 1 = loss + gradients, 2 = materialise cd / loss tensors.
hipError_t dg_launch_corr(const DgCorrArgs& args, int KF, int KD, int nwaves, int mode, hipStream_t stream) {
    const bool simple = args.lo == 0.f && args.hi > 1e30f;   // zero_clamp without stabalize: clamp(cd) == cd * mask
#define DG_CASE(NKF_, NKD_, NW_)                                                                     \
    if (KF == NKF_ * 16 && KD == NKD_ * 16 && nwaves == NW_) {                                       \
        if (simple) {                                                                                \
            if (mode == 1) return launch_corr_t<NKF_, NKD_, NW_, true, false, true>(args, stream);   \
            if (mode == 2) return launch_corr_t<NKF_, NKD_, NW_, false, true, true>(args, stream);   \
            return launch_corr_t<NKF_, NKD_, NW_, false, false, true>(args, stream);                 \
        }                                                                                            \
        if (mode == 1) return launch_corr_t<NKF_, NKD_, NW_, true, false, false>(args, stream);      \
        if (mode == 2) return launch_corr_t<NKF_, NKD_, NW_, false, true, false>(args, stream);      \
        return launch_corr_t<NKF_, NKD_, NW_, false, false, false>(args, stream);                    \
    }
    DG_CASE(8, 6, 4) DG_CASE(8, 6, 8) DG_CASE(8, 8, 4)
    DG_CASE(24, 6, 4) DG_CASE(24, 6, 8) DG_CASE(24, 8, 4)
    DG_CASE(48, 6, 4) DG_CASE(48, 8, 4)
#undef DG_CASE
    return hipErrorInvalidValue;
}

hipError_t dg_launch_finish(const DgFinishArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_corr_finish1, dim3(a.njobs + 1), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_corr_finish2, dim3(1), dim3(64), 0, stream, a);
    return hipGetLastError();
}

// Fused correlation-loss kernel for gfx950 (MI355X).
//
// One workgroup = NWAVES waves (8 = two per SIMD for the ViT-S widths; more than 256 registers per wave makes
// hipcc shuttle MFMA operands between the AGPR and VGPR halves, measured slower); each wave keeps RF x 32
// positions of the stationary operand "R" (normalised feats, bf16) in registers and walks over the streamed
// operand "S" in tiles of 32 positions.  A tile is one contiguous blob in HBM (dg_common.h) that is DMA'd
// into LDS (global_load_lds_dwordx4) two tiles ahead of the computation; one workgroup barrier per tile.
// Per 32x32 tile and row fragment:
//     Yc[s][r] = sum_d Sc[s][d] Rc[r][d]     (KD/16 x v_mfma_f32_32x32x16_f16, fp32 accumulate)
//     Yf[s][r] = sum_k Sf[s][k] Rf[r][k]     (KF/16 x v_mfma_f32_32x32x16_bf16)
//     epilogue (registers only): centering, shift, clamp, loss / cd partial sums, G = dLoss/dcd
//     dR[r][:] += sum_s G[s][r] ScP[s][:]    (accumulator tile reused as the A operand, 2*KD/32 MFMAs)
// With RF = 2 the epilogue VALU work of fragment 0 is interleaved with the MFMA chain of fragment 1.
// The (B,P,P) tensors fd / cd / loss of the reference (src/modules.py:1231-1254) are never written to HBM
// unless a caller asks for them (materialise path).
//
// Reference semantics reproduced here: helper() src/modules.py:1231-1254,
// depth_feature_correlation() :1256-1278 (job kind DG_JOB_DEPTH), norm() :789-790 (backward part).
#include "dg_common.h"
#include <cstdlib>
#include <cstdio>


__device__ __forceinline__ void wait_vmcnt(int n) {   // n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
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
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // over-waits, never under-waits
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

enum { KIND_LANE = 0, KIND_ROW = 1, KIND_DEPTH = 2 };   // centering vector on lanes (R = operand 1) / on tile rows / depth term
typedef int v4i __attribute__((ext_vector_type(4)));
#ifndef DG_STAGGER
#define DG_STAGGER 1
#endif
#ifndef DG_PRIO
#define DG_PRIO 1  // static issue priority: 1 = the half that runs half a tile behind (waves 4-7), 2 = waves 0-3, 0 = none
#endif
#ifndef PF
#define PF 4       // LDS fragment reads kept in flight per wave (4 vs 6 vs 8: 4 is 1-2 % ahead, fewer registers)
#endif

// One accumulator element of a 32x32 tile: (fd, cd) -> loss term and -G = -dLoss/dcd (the sign is folded into the
// constant factors of the backward tail).  vv = per-tile-row value (row mean for KIND_ROW, depth indicator for
// KIND_DEPTH).  KIND_LANE: the accumulator was started at c0_lane, so yf already is fd'' - shift.
// FOLD (gradient pass of the zero_clamp / no stabalize case): the loss and cd sums are not accumulated per element -
// with clamp(cd) = cd * mask they follow from the block's gradient accumulators (corr_body, "partial sums").
template <int KIND, bool SIMPLE, bool FOLD>
__device__ __forceinline__ float epi_elem(float yf, float cdv, float vv, float c0, float nz_lane,
                                          float lo, float hi, float& lsum, float& csum, float& li) {
    float fdv;
    if (KIND == KIND_DEPTH)     fdv = fmaf(nz_lane, vv, c0);
    else if (KIND == KIND_ROW)  fdv = yf + (c0 - vv);
    else                        fdv = yf;
    if (!FOLD) csum += cdv;
    float gneg;
    if (SIMPLE) {                    // zero_clamp, no stabalize: clamp(cd) = cd * mask
        gneg = cdv >= 0.f ? fdv : 0.f;
        if (!FOLD) {
            li = -gneg * cdv;        // = -clamp(cd) * (fd - shift)
            lsum -= li;
        }
    } else {
        const float cl = fminf(fmaxf(cdv, lo), hi);
        lsum = fmaf(cl, fdv, lsum);
        gneg = (cdv >= lo && cdv <= hi) ? fdv : 0.f;
        li = -cl * fdv;
    }
    return gneg;
}

// Final reduction of the per-block partial sums of k_corr_main into the output scalars, by one wave of the next launch
// (DgFinishArgs): per job the partials are summed in double in a fixed (lane-strided, then butterfly) order.
__device__ __forceinline__ void finish_scalars(const DgFinishArgs& f, int lane) {
    double acc[DG_OUT_COUNT];
#pragma unroll
    for (int i = 0; i < DG_OUT_COUNT; ++i) acc[i] = 0.0;
    for (int j = 0; j < f.njobs; ++j) {
        if (!f.part[j]) continue;
        double l = 0.0, c = 0.0;
        const int nb = f.nblk_job[j] ? f.nblk_job[j] : f.nblk;
        for (int i = lane; i < nb; i += 64) { l += f.part[j][2 * i]; c += f.part[j][2 * i + 1]; }
        for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o, 64); c += __shfl_xor(c, o, 64); }
#pragma unroll
        for (int i = 0; i < DG_OUT_COUNT; ++i) {
            if (f.slot_loss[j] == i) acc[i] += -l * (double)f.scale[j];
            if (f.slot_cd[j] == i) acc[i] += c * (double)f.scale[j];
        }
    }
    if (f.nzsum) {        // mean(dd) = mean_n (sum_p nz[n][p])^2 / P^2
        double m = 0.0;
        for (int n = lane; n < f.B; n += 64) { const double s = f.nzsum[n]; m += s * s; }
        for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
        acc[DG_OUT_DD] = m / ((double)f.B * f.P * f.P);
    }
    if (lane == 0) {
        acc[DG_OUT_TOTAL] = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[DG_OUT_TOTAL] += (double)f.wtot[i] * (double)(float)acc[i];
#pragma unroll
        for (int i = 0; i < DG_OUT_COUNT; ++i) f.out[i] = (float)acc[i];
    }
}

__global__ __launch_bounds__(64) void k_finish(const DgFinishArgs f) { finish_scalars(f, threadIdx.x); }

hipError_t dg_launch_finish(const DgFinishArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64), 0, stream, a);
    return hipGetLastError();
}

// NKC: code k-steps of the Y chain actually run (<= NKD): with D <= 16*NKC the remaining fragments are zero padding
template <int NKF, int NKD, int NWAVES, int RF, bool GRAD, bool MAT, bool SIMPLE, int KIND, int NKC = NKD>
__device__ __forceinline__ void corr_body(const DgCorrArgs& args, const DgJob& job, const int n, const int rb, char* smem) {
    using BL = BlobT<NKF, NKD>;
    constexpr int KD = BL::KD;
    constexpr int NDF = KD / 32;            // 32-wide output fragments of dR
    constexpr int DP = KD;                  // padded code width of dR
    constexpr int BUF = BL::BYTES + 256;    // blob + 32 per-row floats (+ copy)
    constexpr int RCB = BL::OFF_P - BL::OFF_C;   // bytes of one C part (the stationary code rows of one fragment)
    constexpr int NBUF = BL::BYTES > 48 * 1024 ? 2 : 3;   // LDS buffers; tiles are fetched NBUF-1 ahead
    constexpr bool RCREG = BL::BYTES > 48 * 1024;         // LDS full (ViT-B): stationary code rows live in registers
    constexpr int NSF = KIND == KIND_DEPTH ? 0 : NKF;     // feature k-steps per tile
    constexpr int NS = NKC + NSF;                         // MFMA steps of one Y chain
    static_assert(!RCREG || RF == 1, "register-resident code rows only with one fragment per wave");
    constexpr bool GOUT = GRAD && KIND == KIND_LANE;     // pass-A helper jobs store their G tiles for k_gs
    constexpr bool STAG = DG_STAGGER && NWAVES == 8 && RF == 1 && NBUF == 3 && !MAT;
    // loss / cd sums from the gradient accumulators (see epi_elem).  Not for the depth term: its G takes two or three
    // distinct values, so their fp16 rounding would be a bias (2e-4 relative) instead of noise.
    constexpr bool FOLD = GRAD && SIMPLE && !MAT && KIND != KIND_DEPTH;

    // kernarg fields used inside the tile loop are copied to locals: behind the asm memory clobbers hipcc would re-load
    // them (s_load + s_waitcnt lgkmcnt(0), which also drains the LDS read ring) in every iteration
    const int dbg = DG_DBG(args.debug);
    uint16_t* const Gout = job.Gout;
    float* const out_cd = job.out_cd;
    float* const out_loss = job.out_loss;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int Ppad = args.Ppad, P = args.P;
    const int ntiles_all = Ppad >> 5;
    // developer timing stamps (DG_STAMPS=<file>): phase times of one block, kept in LDS until the end
    uint32_t* const st_lds = reinterpret_cast<uint32_t*>(smem + NBUF * BUF + (RCREG ? 0 : NWAVES * RF * RCB) + NWAVES * 2 * 4);
#ifdef DG_STAMP_BUILD      // make EXTRA=-DDG_STAMP_BUILD: instrumented build (perturbs the schedule slightly)
    const bool stamping = args.stamps != nullptr && n == 0 && rb == 0 && KIND == KIND_LANE;
    auto STAMP = [&](int t, int ph) {
        if (stamping && lane == 0 && t < 25) st_lds[(wid * 25 + t) * 4 + ph] = (uint32_t)__builtin_readcyclecounter();
    };
    auto BLOG = [&](int k) {
        if (args.blocklog && tid == 0) {
            unsigned long long* e = args.blocklog + (size_t)blockIdx.x * 8;
            if (k == 0) { e[0] = __builtin_amdgcn_s_getreg(63492); e[1] = __builtin_amdgcn_s_getreg(63508); e[6] = KIND; e[7] = rb; }
            e[2 + k] = wall_clock64();
        }
    };
#else
    constexpr bool stamping = false;
    auto STAMP = [&](int, int) {};
    auto BLOG = [&](int) {};
    (void)st_lds;
#endif
    BLOG(0);
    const int ntiles = (dbg & 128) ? 1 : ntiles_all;     // developer ablation: one tile only (fixed per-block cost)
    const int nR = job.ridx ? (int)job.ridx[n] : n;
    const int nS = job.sidx ? (int)job.sidx[n] : n;

    // ---- ragged last row block with ONE row tile (e.g. P = 784: 24.5 tiles): waves 0 and 1 both take that tile and share
    //      the S tiles (even / odd), the other waves fetch; four tile buffers (the unused code-row slots hold the fourth).
    //      Without this the block runs one wave for as long as a full block runs eight.
    const int nact = min(NWAVES, (ntiles_all - rb * NWAVES * RF + RF - 1) / RF);
    const bool pair = DG_STAGGER && NWAVES == 8 && RF == 1 && NBUF == 3 && !MAT && !RCREG && nact == 1 && ntiles > 1 &&
                      !(dbg & 67108864);
    const bool owner = !(pair && wid == 1);               // wave 1 hands its gradient accumulators to wave 0 at the end
    // ---- the RF 32-row tiles of R owned by this wave
    const int rtile0 = (rb * NWAVES + ((pair && wid == 1) ? 0 : wid)) * RF;
    bool act[RF];
    int pr[RF];
    const char* Rblob[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) {
        act[f] = rtile0 + f < ntiles_all;                       // wave-uniform
        pr[f] = act[f] ? (rtile0 + f) * 32 + r : 0;         // stationary position of this lane (clamped when idle)
        Rblob[f] = job.Rop + ((size_t)nR * ntiles_all + (act[f] ? rtile0 + f : 0)) * BL::BYTES;
    }
    const bool wave_active = act[0];
    // global stores this wave issues per tile (G tiles; younger than the tile DMAs, counted by vmcnt like them)
    const int nst = GOUT ? 2 * ((act[0] ? 1 : 0) + (RF == 2 && act[RF - 1] ? 1 : 0)) : 0;

    // ---- stationary operand: feats fragments -> registers, code rows -> LDS (DMA of the C part of its blob)
    const uint32_t smem_a = lds_addr(smem);
    char* rc_lds = smem + NBUF * BUF + (pair ? 0 : wid) * (RF * RCB);
    f16x8 Rc[RCREG ? NKD : 1];
    if (RCREG) {
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) {
            Rc[ks] = *reinterpret_cast<const f16x8*>(Rblob[0] + BL::OFF_C + ((2 * ks + h) * 32 + r) * 16);
            asm volatile("" : "+v"(Rc[ks]));
        }
    } else if (!pair || wid == 0) {
#pragma unroll
        for (int f = 0; f < RF; ++f)
            for (int c = 0; c < RCB / 1024; ++c)
                dma16(Rblob[f] + BL::OFF_C + c * 1024 + lane * 16, smem_a + NBUF * BUF + (wid * RF + f) * RCB + c * 1024);
    }
    // exact clamp masks (k_cd_mask; small sample grids, one wave per SIMD: the words of this lane's R position for all <= 8 S tiles)
    constexpr bool XMASK = GRAD && SIMPLE && !MAT && KIND == KIND_LANE && NWAVES == 4 && RF == 1;
    uint32_t mw[XMASK ? 8 : 1];
    const bool xmask = XMASK && job.maskbits != nullptr && ntiles_all <= 8;
    if constexpr (XMASK) {
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) mw[t8] = 0u;
        if (xmask) {            // (one uniform branch, clamped tile indices: guarded loads are issued and waited for one by one)
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) mw[t8] = job.maskbits[((size_t)n * ntiles_all + min(t8, ntiles_all - 1)) * Ppad + pr[0]];
        }
    }
    bf16x8 Rf[RF][NSF > 0 ? NSF : 1];
    if (KIND != KIND_DEPTH && !(dbg & 2048)) {
#pragma unroll
        for (int f = 0; f < RF; ++f)
#pragma unroll
            for (int ks = 0; ks < NKF; ++ks)
                Rf[f][ks] = *reinterpret_cast<const bf16x8*>(Rblob[f] + dg_f_off(r, 2 * ks + h));
    }
    // Make hipcc wait for the fragment loads at ONE place (right after the first tile DMAs are issued, so the two
    // latencies overlap): its counted vmcnt waits at their first use inside the tile loop would otherwise also count
    // (and drain) the tile DMAs it does not know about.
    auto settle_R = [&]() {
        if constexpr (XMASK) {
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) asm volatile("" : "+v"(mw[t8]));
        }
        if (KIND != KIND_DEPTH && !(dbg & 1024)) {
#pragma unroll
            for (int f = 0; f < RF; ++f)
#pragma unroll
                for (int ks = 0; ks < NKF; ++ks) asm volatile("" : "+v"(Rf[f][ks]));
        }
    };

    // ---- per-job scalars
    float c0 = -job.shift;    // fd'' - shift = Yf - rowmean + (m0 - shift)
    if (KIND != KIND_DEPTH && job.rvec) {       // m0: B per-image sums, added in a fixed order by every wave
        float m = 0.f;
        for (int i = lane; i < args.B; i += 64) m += job.rimg[i];
        c0 += wave_sum(m) * args.inv_BP;
    }
    float c0_lane[RF], nz_lane[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) {
        c0_lane[f] = c0; nz_lane[f] = 0.f;
        if (KIND == KIND_LANE && job.rvec) c0_lane[f] -= job.rvec[(size_t)n * Ppad + pr[f]];
        if (KIND == KIND_DEPTH) nz_lane[f] = job.nzR[(size_t)n * Ppad + pr[f]];
    }
    const float lo = args.lo, hi = args.hi;
    const bool has_vec = KIND == KIND_DEPTH || (KIND == KIND_ROW && job.rvec != nullptr);

    // ---- tile staging by LDS-DMA (1 KiB per wave instruction, linear in HBM and in LDS)
    const char* Sbase = job.Sop + (size_t)nS * ntiles_all * BL::BYTES + lane * 16;
    const float* vsrc = reinterpret_cast<const float*>(args.dummy);      // per-row vector of the tile (rvec or nz)
    if (KIND == KIND_ROW && job.rvec) vsrc = job.rvec + (size_t)n * Ppad;
    if (KIND == KIND_DEPTH) vsrc = job.nzS + (size_t)n * Ppad;
    constexpr int C_BEGIN = KIND == KIND_DEPTH ? BL::CHUNK_C0 : 0;
    constexpr int C_END = GRAD ? BL::CHUNKS : BL::CHUNK_P0;
    constexpr int PIECES = (C_END - C_BEGIN + NWAVES - 1) / NWAVES;
    // which 1-KiB chunks of a tile this wave fetches: normally chunk c belongs to wave NWAVES-1 - c % NWAVES; in a ragged
    // last row block (fewer than half of the waves own rows) the idle waves share all chunks
    const int ncomp = pair ? 2 : nact;                      // waves that compute in this block
    const bool idle_fetch = RF == 1 && nact > 0 && nact < NWAVES / 2 && !(dbg & 33554432);
    const int dma_stride = idle_fetch ? NWAVES - ncomp : NWAVES;
    const int dma_first = idle_fetch ? (wid >= ncomp ? wid - ncomp : -1) : 0;
    auto issue = [&](int t, int b) {
        const char* src = Sbase + (size_t)t * BL::BYTES;
        // (buffer 3: pair mode only, behind wave 0's code rows)
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_a + b * BUF + (b == 3 ? RCB : 0));
        // chunk c goes to wave NWAVES-1 - c % NWAVES (a ragged remainder lands on the waves that run half a tile behind);
        // the counted waits below use this wave's own number of pieces (my_dma)
        if (dma_stride == NWAVES) {
#pragma unroll
            for (int k = 0; k < PIECES; ++k) {
                const int c = C_BEGIN + (NWAVES - 1 - wid) + k * NWAVES;
                if (c < C_END) dma16(src + c * 1024, dst + c * 1024);
            }
        } else if (dma_first >= 0) {     // ragged last row block: the waves without rows fetch the tiles for the few that have
            for (int c = C_BEGIN + dma_first; c < C_END; c += dma_stride) dma16(src + c * 1024, dst + c * 1024);
        }
        if (KIND != KIND_LANE)     // 32 floats of the tile rows; all waves write the same bytes (lanes 32-63: a copy behind)
            dma4(vsrc + t * 32 + (lane & 31), dst + BL::BYTES);
    };
    const int my_dma = (idle_fetch ? (dma_first >= 0 ? (C_END - C_BEGIN - dma_first + dma_stride - 1) / dma_stride : 0)
                                   : (C_END - C_BEGIN - (NWAVES - 1 - wid) + NWAVES - 1) / NWAVES) +
                       (KIND != KIND_LANE ? 1 : 0);   // DMA instructions of this wave per tile

    f32x16 dR[RF][NDF];
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int d = 0; d < NDF; ++d) dR[f][d] = f32x16{};
    float lsum = 0.f, csum = 0.f;

    // A fragment of feature k-step st = granule 2st+h of tile row r: dg_f_off(r, 2st+h) = per-lane base[st % FPER] + constant
    constexpr int FPER = DG_F_IG >= 2 ? DG_F_IG / 2 : 1;      // k-steps per interleave group
    int fbase[FPER];
#pragma unroll
    for (int j = 0; j < FPER; ++j) fbase[j] = dg_f_off(r, 2 * j + h);
    const int crow = (h * 32 + r) * 16;

    f32x16 Yf[RF], Yc[RF];       // live across the barrier in the staggered schedule

    // ---- one Y chain = NSF feature k-steps then NKD code k-steps.  The A operands (S fragments) run through an
    //      explicit ring of PF registers that is refilled right after each MFMA, so that PF LDS reads are always in
    //      flight; scheduling fences pin that order (left alone hipcc serialises read -> wait -> MFMA here).
    const int late_prio = (STAG && DG_PRIO == 1 && wid >= NWAVES / 2 && !(dbg & 32768)) ? 1 : 0;   // priority outside the chain
    auto chain = [&](const char* tile, const int f, auto&& between) {
        auto a_ptr = [&](int st) -> const v4i* {
            return st < NSF ? reinterpret_cast<const v4i*>(tile + fbase[st % FPER] + (st / FPER) * (FPER * 1024))
                            : reinterpret_cast<const v4i*>(tile + BL::OFF_C + crow + (st - NSF) * 1024);
        };
        Yc[f] = f32x16{};
        if (KIND == KIND_LANE) {          // start the feature accumulator at c0_lane: fd'' - shift comes out of the chain
#pragma unroll
            for (int i = 0; i < 16; ++i) Yf[f][i] = c0_lane[f];
        } else {
            Yf[f] = f32x16{};
        }
        v4i ra[PF], rb[2];
#pragma unroll
        for (int i = 0; i < PF; ++i) if (i < NS) ra[i] = *a_ptr(i);
        if (!RCREG) {
#pragma unroll
            for (int k = 0; k < 2; ++k) rb[k] = *reinterpret_cast<const v4i*>(rc_lds + f * RCB + crow + k * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the wave in its MFMA chain wins the issue port: an MFMA needs it 8 cycles in 32, the partner's epilogue VALU
        // stream would otherwise starve the matrix pipe (stamps: iteration 4410 -> 4070 cycles)
        if (STAG && !(dbg & 131072)) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const v4i cur = ra[st % PF];
            if (st < NSF) {
                Yf[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur), Rf[f][st < NSF ? st : 0], Yf[f], 0, 0, 0);
            } else {
                const int k = st - NSF;
                const f16x8 b = RCREG ? Rc[k] : __builtin_bit_cast(f16x8, rb[k & 1]);
                Yc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur), b, Yc[f], 0, 0, 0);
                if (!RCREG && k + 2 < NKC) rb[k & 1] = *reinterpret_cast<const v4i*>(rc_lds + f * RCB + crow + (k + 2) * 1024);
            }
            if (st + PF < NS) ra[st % PF] = *a_ptr(st + PF);
            between(st);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STAG && !(dbg & 131072)) { if (late_prio) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    };

    // ---- everything after the Y chains of tile t: epilogue (VALU) and the gradient MFMAs.  For RF == 2 the epilogue of
    //      fragment 0 is carried in the gaps of fragment 1's chain (chain1 = true).
    auto post = [&](const char* tile, const int t, const bool chain1) {
        // per-tile-row vector (rows 8*i4 + 4*h + e): 4 x 16-byte LDS reads, shared by the fragments
        float vv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) vv[i] = 0.f;
        if (KIND != KIND_LANE && has_vec) {
            const float* rvs = reinterpret_cast<const float*>(tile + BL::BYTES);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const float4 v4 = *reinterpret_cast<const float4*>(rvs + 8 * i4 + 4 * h);
                vv[4 * i4] = v4.x; vv[4 * i4 + 1] = v4.y; vv[4 * i4 + 2] = v4.z; vv[4 * i4 + 3] = v4.w;
            }
        }
        f16x8 ga[RF][2];                   // G as the A operand of the gradient product: k-step sp holds elements 8sp..8sp+7
        auto epi = [&](int f, int i) {     // accumulator element i = (tile row (i&3)+8*(i>>2)+4*h, lane column r)
            float li;
            float cdm = Yc[f][i];
            if constexpr (XMASK) {
                // the exact sign of cd instead of the fp16-operand one (same magnitude: the value only enters the loss sums)
                uint32_t w8 = mw[0];
#pragma unroll
                for (int t8 = 1; t8 < 8; ++t8) w8 = t == t8 ? mw[t8] : w8;
                const bool on = (w8 >> ((i & 3) + 8 * (i >> 2) + 4 * h)) & 1u;
                if (xmask) cdm = on ? fabsf(cdm) : -fabsf(cdm) - 1e-30f;
            }
            ga[f][i >> 3][i & 7] = (_Float16)epi_elem<KIND, SIMPLE, FOLD>(Yf[f][i], cdm, vv[i], c0, nz_lane[f], lo, hi, lsum, csum, li);
            if (MAT) {   // R = operand 2 on lanes -> stores contiguous along q
                const int p = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (act[f] && pr[f] < P && p < P) {
                    // (identity grid: positions are pixel indices; the reference's tensors are indexed by x*S + y)
                    const int w_ = args.pos_w;
                    const int po = w_ ? (p % w_) * w_ + p / w_ : p, qo = w_ ? (pr[f] % w_) * w_ + pr[f] / w_ : pr[f];
                    const size_t o = ((size_t)n * P + po) * P + qo;
                    if (out_cd) out_cd[o] = KIND == KIND_DEPTH ? nz_lane[f] * vv[i] : Yc[f][i];
                    if (out_loss) out_loss[o] = li;
                }
            }
        };
        if (RF == 2 && chain1) {
            chain(tile, RF - 1, [&](int st) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (i * NS / 16 == st) epi(0, i);
            });
#pragma unroll
            for (int i = 0; i < 16; ++i) epi(RF - 1, i);
        } else if (dbg & 16) {   // developer ablation: no epilogue VALU (results invalid)
#pragma unroll
            for (int f = 0; f < RF; ++f) { ga[f][0] = __builtin_bit_cast(f16x8, Yf[f][0] > 1e30f ? v4i{1, 1, 1, 1} : v4i{0, 0, 0, 0}); ga[f][1] = ga[f][0]; lsum += Yc[f][0]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) epi(0, i);
            if (RF == 2) {      // idle second fragment (ragged last row block): contributes nothing
                ga[RF - 1][0] = f16x8{}; ga[RF - 1][1] = f16x8{};
            }
        }
        if (GRAD && GOUT && !(dbg & 8)) {
            // G tile (fp16, accumulator order) -> HBM, 2 KiB per wave and tile, fully coalesced: input of k_gs, which
            // forms the gradient of the STREAMED operand's code without recomputing fd / cd
#pragma unroll
            for (int f = 0; f < RF; ++f)
                if (act[f]) {
                    // non-temporal: the tiles are read by the NEXT kernel only; keeping them out of the way leaves the L2 to
                    // the operand blobs this kernel streams over and over
                    // tile = [2 k-steps][64 lanes][16 B]: every store instruction writes one contiguous KiB
                    v4i* g = reinterpret_cast<v4i*>(Gout) + (((size_t)n * ntiles_all + t) * ntiles_all + rtile0 + f) * 128 + lane;     // [image][S tile][R tile]
                    __builtin_nontemporal_store(__builtin_bit_cast(v4i, ga[f][0]), g);
                    __builtin_nontemporal_store(__builtin_bit_cast(v4i, ga[f][1]), g + 64);
                }
        }
        if (GRAD && !(dbg & 4)) {
            // dR[f][r][:] += sum_s G[f][s][r] * ScP[s][:]   (accumulator tile as the A operand; B shared by the fragments)
            // k-step outer, channel group inner: consecutive MFMAs go to different accumulators
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int d = 0; d < NDF; ++d) {
                    const f16x8 b = *reinterpret_cast<const f16x8*>(tile + BL::OFF_P + (h * KD + 32 * d + r) * 16 + sp * (2 * KD * 16));
#pragma unroll
                    for (int f = 0; f < RF; ++f)
                        dR[f][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[f][sp], b, dR[f][d], 0, 0, 0);
                }
        }
    };

    if (STAG) {
        // Staggered schedule (two waves per SIMD): waves NWAVES/2.. run half a tile behind their SIMD partners - they do
        // the epilogue + gradient MFMAs of tile t-1 first and then the Y chain of tile t - so that one wave's VALU
        // epilogue runs under the other's MFMA chain instead of both alternating in lockstep behind the tile barrier.
        // Tile t-1 must stay in LDS during iteration t, so tiles are fetched one (not two) ahead.
        const bool late = wid >= NWAVES / 2;
        {   // static priority (experiment bits: 32768 = none, 65536 = waves 0-3 instead)
            const int prio_mode = (dbg & 32768) ? 0 : ((dbg & 65536) ? 2 : DG_PRIO);
            if (prio_mode == 1 && late) __builtin_amdgcn_s_setprio(1);
            if (prio_mode == 2 && !late) __builtin_amdgcn_s_setprio(1);
        }
        // pair mode (ragged last row block): every wave takes the "early" schedule; waves 0 and 1 compute the even / odd tiles
        // of a pair per barrier, waves 2-7 fetch the next pair into the other two of four buffers
        const bool late2 = late && !pair;
        auto tilep = [&](int t) -> const char* {
            const int b = pair ? (t & 3) : t % 3;
            return smem + b * BUF + (b == 3 ? RCB : 0);
        };
        issue(0, 0);
        if (pair) issue(1, 1);
        settle_R();
        auto top = [&](int t) {          // same barrier sequence in both halves
            STAMP(t, 0);
            // tile t landed; only the G stores of the previous tile may still be in flight (fetch-only waves: nothing may)
            wait_vmcnt((pair && dma_first >= 0) ? 0 : nst);
            if (!(dbg & 256)) __builtin_amdgcn_s_barrier();
            STAMP(t, 1);
            if (t == 0) BLOG(1);
            if (pair) {
                if (!(dbg & 1)) {
                    if (t + 2 < ntiles) issue(t + 2, (t + 2) & 3);
                    if (t + 3 < ntiles) issue(t + 3, (t + 3) & 3);
                }
            } else {
                // waves 4-7 fetch the next tile here; waves 0-3 do it after their chain (the issue of 4-5 LDS-DMA pieces costs
                // 400-600 cycles in front of the chain, but hides beside the partner's chain) - bit 524288 switches that off
                if (t + 1 < ntiles && !(dbg & 1) && (late2 || (dbg & 524288))) issue(t + 1, (t + 1) % 3);
            }
            STAMP(t, 2);
        };
        if (!late2) {
            const int tstep = pair ? 2 : 1, tfirst = pair ? wid : 0;
            for (int t0 = 0; t0 < ntiles; t0 += tstep) {
                top(t0);
                const int t = t0 + tfirst;
                const char* tile = tilep(t);
                const bool work = wave_active && t < ntiles;
                if (work) { chain(tile, 0, [](int) {}); STAMP(t0, 3); }
                if (!pair && t0 + 1 < ntiles && !(dbg & 1) && !(dbg & 524288)) issue(t0 + 1, (t0 + 1) % 3);
                if (work) post(tile, t, false);
            }
        } else {
            top(0);
            if (wave_active) chain(smem, 0, [](int) {});
            for (int t = 1; t < ntiles; ++t) {
                top(t);
                if (wave_active) {
                    post(smem + ((t - 1) % 3) * BUF, t - 1, false);
                    STAMP(t, 3);
                    chain(smem + (t % 3) * BUF, 0, [](int) {});
                }
            }
            if (wave_active) post(smem + ((ntiles - 1) % 3) * BUF, ntiles - 1, false);
        }
    } else {
        issue(0, 0);
        if (NBUF == 3 && ntiles > 1) issue(1, 1);
        settle_R();
        int bcur = 0;
        for (int t = 0; t < ntiles; ++t) {
            // tile t has landed (this wave's pieces: vmcnt; everybody's: barrier); the buffer of tile t-1 is free again
            wait_vmcnt((NBUF == 3 && t + 1 < ntiles) ? my_dma + nst : nst);
            __builtin_amdgcn_s_barrier();
            if (t + NBUF - 1 < ntiles && !(dbg & 1)) issue(t + NBUF - 1, bcur >= 1 ? bcur - 1 : NBUF - 1);
            const char* tile = smem + bcur * BUF;
            bcur = bcur == NBUF - 1 ? 0 : bcur + 1;
            if (!wave_active) continue;
            chain(tile, 0, [](int) {});
            post(tile, t, RF == 2 && act[RF - 1]);
        }
    }

    // ---- store the RAW gradient w.r.t. the normalised stationary code in accumulator order (first: the stores drain
    //      while the block sums are formed and published)
    //      [image][R tile][channel group d][i>>2][lane][i&3] (16 bytes per lane and store, 1 KiB per wave instruction).
    //      The normalisation backward is linear with the same x for every pair-set whose stationary operand is operand 1,
    //      so k_grad_combine applies it once to the weighted sum of these buffers.
    BLOG(2);
    if (pair && GRAD) {          // wave 1's share of the gradient accumulators -> wave 0 (through the free tile buffers)
        __syncthreads();
        float* xch = reinterpret_cast<float*>(smem);
        if (wid == 1) {
#pragma unroll
            for (int d = 0; d < NDF; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) xch[(d * 16 + i) * 64 + lane] = dR[0][d][i];
        }
        __syncthreads();
        if (wid == 0) {
#pragma unroll
            for (int d = 0; d < NDF; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) dR[0][d][i] += xch[(d * 16 + i) * 64 + lane];
        }
    }
    if (GRAD && job.dR && owner && !(dbg & 512)) {
#pragma unroll
        for (int f = 0; f < RF; ++f) {
            if (!act[f]) continue;
            float* base = job.dR + ((size_t)n * (Ppad >> 5) + rtile0 + f) * (32 * DP) + lane * 4;
#pragma unroll
            for (int d = 0; d < NDF; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {dR[f][d][4 * g], dR[f][d][4 * g + 1], dR[f][d][4 * g + 2], dR[f][d][4 * g + 3]};
                    if (32 * d + r < args.D)      // padding channels (D..KD-1) are never read: do not spend HBM writes on them
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + (d * 4 + g) * 256));
                }
        }
    }

    // ---- partial sums of this block (deterministic two-level reduction; finished by the last block to retire)
    // (rows of an inactive second fragment are all-zero operands? no: they are clamped copies -> excluded here)
    float* red = reinterpret_cast<float*>(smem + NBUF * BUF + (RCREG ? 0 : NWAVES * RF * RCB));
    if (FOLD) {
        // clamp(cd) = cd * mask, so with G' = mask * (fd'' - shift):
        //   sum_pq clamp(cd) (fd'' - shift) = sum_pq G'_pq <x_p, y_q> = sum_p <x_p, dR'_p>      (dR' = the gradient accumulators)
        //   sum_pq cd_pq                    = sum_p <x_p, sum_q y_q>                              (column sums of the streamed code)
        // x_p = this wave's stationary code rows (LDS or registers), dR[f][d][i] = (row (i&3)+8*(i>>2)+4*h, channel 32 d + r)
        lsum = 0.f; csum = 0.f;
#pragma unroll
        for (int f = 0; f < RF; ++f) {
            if (!act[f] || !owner) continue;
            float cs[NDF];
#pragma unroll
            for (int d = 0; d < NDF; ++d) cs[d] = (KIND != KIND_DEPTH && job.Scsum) ? job.Scsum[(size_t)nS * KD + 32 * d + r] : 0.f;
            // x in the layout of dR (rows in registers, channel on the lane) = X * I: the stationary code fragments (the
            // B operands of the cd chain, same register layout as an A fragment) times 0/1 selector fragments
            f16x8 sel[2];
#pragma unroll
            for (int sI = 0; sI < 2; ++sI)
#pragma unroll
                for (int j = 0; j < 8; ++j) sel[sI][j] = (8 * h + j + 16 * sI == r) ? (_Float16)1.f : (_Float16)0.f;
#pragma unroll
            for (int d = 0; d < NDF; ++d) {
                f32x16 X = f32x16{};
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    const int k = 2 * d + sI;
                    const f16x8 xa = RCREG ? Rc[k] : *reinterpret_cast<const f16x8*>(rc_lds + f * RCB + crow + k * 1024);
                    X = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, sel[sI], X, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    lsum = fmaf(X[i], dR[f][d][i], lsum);
                    csum = fmaf(X[i], cs[d], csum);
                }
            }
        }
    }
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

    BLOG(3);
    if (stamping) {
        __syncthreads();
        for (int i = tid; i < NWAVES * 100; i += NWAVES * 64) args.stamps[i] = st_lds[i];
    }
}

template <int NKF, int NKD, int NWAVES, int RF, bool GRAD, bool MAT, bool SIMPLE, int NKC = NKD>
__global__ __launch_bounds__(NWAVES * 64) void k_corr_main(const DgCorrArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][BUF] + Rc[NWAVES][RF][C part] + red[NWAVES][2]
    // ---- XCD-aware block order: blocks that share an XCD (orig % 8) get a contiguous range of logical ids,
    //      image-major inside: all pair-sets of one image (which stream the same operand blobs) meet in one L2.
    int bid;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    }
    const int per_img = args.njobs * args.nrb;
    int n, jid, rb;
    const int nd = args.jobs[args.njobs - 1].kind == DG_JOB_DEPTH ? 1 : 0, nh = args.njobs - nd;
    if ((gridDim.x & 7) == 0 && (args.B & 7) == 0 && nh > 0 && !(DG_DBG(args.debug) & 8388608)) {
        // every XCD owns B/8 whole images; inside that chunk the long blocks go first (longest-processing-time order):
        // pair-set jobs with a full row block, then their ragged last row block, then the cheap depth job
        const int imgs = args.B >> 3, per_chunk = imgs * per_img;
        const int xcd = bid / per_chunk;
        int i = bid - xcd * per_chunk;
        const bool ragged = args.nrb > 1 && ((args.Ppad >> 5) % (NWAVES * RF)) != 0;
        const int nfull = args.nrb - (ragged ? 1 : 0);
        const int cA = imgs * nh * nfull, cB = ragged ? imgs * nh : 0;
        int nl;
        if (i < cA) {
            nl = i / (nh * nfull); i -= nl * nh * nfull; jid = i / nfull; rb = i - jid * nfull;
        } else if (i < cA + cB) {
            i -= cA; nl = i / nh; jid = i - nl * nh; rb = args.nrb - 1;
        } else {
            i -= cA + cB; nl = i / args.nrb; jid = nh; rb = i - nl * args.nrb;
        }
        n = xcd * imgs + nl;
    } else {
        n = bid / per_img;
        bid -= n * per_img;
        jid = bid / args.nrb;
        rb = bid - jid * args.nrb;
    }
    const DgJob& job = args.jobs[jid];
    if ((DG_DBG(args.debug) & 16777216) && rb == args.nrb - 1 && args.nrb > 1) return;     // (ablation: skip the last row block)
    if ((DG_DBG(args.debug) & 134217728) && job.kind == DG_JOB_DEPTH) return;               // (ablation: skip the depth job)
    __shared__ unsigned long long span_keep[2];
    if (!MAT && threadIdx.x == 0) dg_span_enter(args.span, span_keep);
    if (job.kind == DG_JOB_DEPTH) corr_body<NKF, NKD, NWAVES, RF, GRAD, MAT, SIMPLE, KIND_DEPTH, NKC>(args, job, n, rb, smem);
    else if (job.center_on_lane == 0) corr_body<NKF, NKD, NWAVES, RF, GRAD, MAT, SIMPLE, KIND_ROW, NKC>(args, job, n, rb, smem);
    else if (!MAT) corr_body<NKF, NKD, NWAVES, RF, GRAD, MAT, SIMPLE, KIND_LANE, NKC>(args, job, n, rb, smem);
    if (!MAT && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dg_span_exit(args.span, span_keep); }
}

// ---- k_gs: gradient w.r.t. the streamed operand's code from the stored G tiles ---------------------------------
//     dS[q][:] = sum_p G[q][p] * x_R[p][:]     per image and pair-set: (P x P) fp16 G, read exactly once from HBM
// One block = GS_CW consumer waves + 1 producer wave.  Consumer wave c owns S tile blockIdx.x*GS_CW + c (32 positions q)
// and walks over the R tiles (the K dimension); per R tile it needs
//   * its own G tile - the producing wave's accumulator-order fp16 registers (lane = R position p, 16 S positions per
//     lane), 2 KiB, streamed from HBM (non-temporal) three tiles ahead into a register ring;
//   * the P part of the R operand's blob (the B fragments: granule c of channel d = 8 positions in dg_perm32 order),
//     shared by all consumers: the producer wave moves it global -> LDS with the LDS-DMA into a ring of GS_NB buffers,
//     GS_NB-1 tiles ahead, and is the only wave that waits on it (counted vmcnt), so the one barrier per R tile never
//     exposes a load latency.
// The G tile is transposed through a per-wave LDS scratch: four 8-byte stores per lane into a swizzled [p][q] image, read
// back with the transposing ds_read_b64_tr_b16 (rows picked in the position order of the P part's granules); 6 MFMAs
// accumulate it.  Then the normalisation backward with the
// S code is applied in registers and the accumulator tile is written as a gradient tile (dg_gtile_off).
// grid (ceil(nt/GS_CW), B, jobs), block (GS_CW+1)*64, dynamic LDS GS_NB P parts + GS_CW scratch tiles.
#ifndef GS_CW
#define GS_CW 7
#endif
#ifndef GS_NB
#define GS_NB 6
#endif
#define GS_TS 80            // row stride (bytes) of the transposition scratch
template <int CH>
__device__ __forceinline__ void gs_wait_tiles(int k) {   // at most k tiles (CH DMA instructions each) still in flight
    switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CH) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * CH) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * CH) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * CH) : "memory"); break;
    }
}
// ---- the depth term as blocks of the k_gs launch (depth_feature_correlation, src/modules.py:1256-1278) -------------------------
// loss = -clamp(cd)(dd - shift), cd = corr(code, code) of the image itself, dd[p][q] = nz_p nz_q (quirk Q1); by symmetry only the
// stationary-side gradient is formed (the combine doubles it).  One block = 8 row tiles of one image, one 32-row fragment per
// wave (the whole block lives in k_gs's 128 registers per wave and its 54 KB of LDS, so it runs beside the G-stream blocks):
// the streamed tile's C and P parts + its 32 indicators come through two LDS buffers, register-staged one tile ahead (plain
// loads: with eight waves per block and other blocks on the CU there is enough in flight); per tile NKD cd MFMAs, the epilogue
// with per-element sums (G takes two or three distinct values: the fold of the helper jobs would bias), 6 gradient MFMAs.
template <int NKF, int NKD, bool XM>      // XM: exact clamp masks of the intra pair-set (k_cd_mask), dep_maskbits
__device__ __forceinline__ void gs_depth_block(const DgGsArgs& a, const uint32_t* dep_maskbits, const int dbid, char* smem) {
    using BL = BlobT<NKF, NKD>;
    constexpr int KD = BL::KD, NDF = KD / 32, NTH = (GS_CW + 1) * 64;
    constexpr int TILE = 256 + (BL::BYTES - BL::OFF_C);       // [32 indicators, padded][C part][P part]
    constexpr int NPC = (BL::BYTES - BL::OFF_C) / 16;          // 16-byte pieces of the C and P parts
    static_assert(NPC <= 2 * NTH && 2 * TILE <= GS_NB * (BL::BYTES - BL::OFF_P), "two tiles fit the ring; two pieces per thread");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    const int n = dbid / a.dep_nrb, rb = dbid - n * a.dep_nrb;
    const int ntiles = a.Ppad >> 5;
    const int rtile = rb * (GS_CW + 1) + wid;
    const bool act = rtile < ntiles;
    const char* const img = a.dep_op + (size_t)n * ntiles * BL::BYTES;
    const float* const nz = a.dep_nz + (size_t)n * a.Ppad;
    // stationary fragment: code rows of this wave's tile as B operands, its indicators per lane
    v4i Rc[NKD];
    const char* Rb = img + (size_t)(act ? rtile : 0) * BL::BYTES + BL::OFF_C;
#pragma unroll
    for (int k = 0; k < NKD; ++k) Rc[k] = *reinterpret_cast<const v4i*>(Rb + ((2 * k + h) * 32 + r) * 16);
    // (k-steps that are all channel padding: zero HERE - with k_corr2's FOLD the operand-1 blobs carry the intra row means in
    //  the first such k-step of the C part, dg_corr2.hip; a zero stationary fragment makes the product blind to it)
#pragma unroll
    for (int k = 0; k < NKD; ++k) if (16 * k >= a.D) Rc[k] = v4i{0, 0, 0, 0};
    const float nz_lane = act ? nz[rtile * 32 + r] : 0.f;
    const float c0 = -a.dep_shift;
    f32x16 acc[NDF];
#pragma unroll
    for (int d = 0; d < NDF; ++d) acc[d] = f32x16{};
    float lsum = 0.f;
    // tile staging: thread t owns pieces t and t + NTH of the C/P parts, threads 0..7 also four indicators each
    v4i st0, st1 = v4i{0, 0, 0, 0};
    f32x4 stz = f32x4{0.f, 0.f, 0.f, 0.f};
    // XM - exact clamp masks of the intra pair-set (zero_clamp, no upper bound; <= 8 tiles): the words of this lane's R position
    // against every S tile, loaded up front and waited for once.  The block without them sits at k_gs's 128 registers (9 spilled);
    // eight more spill 80 (KD = 96) / 454 (KD = 128) and cost 11 - 27 us at C3's shape, so the masked form is its own
    // instantiation of k_gs with 256 registers per wave (one block per CU: small sample grids have few blocks).
    // Dense grids with DG_EXACT_MASKS (more than eight tiles): the word of tile t + 1 is requested while tile t is worked on.
    uint32_t mw[XM ? 8 : 1];
    uint32_t mw_next = 0u;
    const bool mw_pre = ntiles <= 8;
    const uint32_t* const mw_row = XM ? dep_maskbits + (size_t)n * ntiles * a.Ppad + (act ? rtile : 0) * 32 + r : nullptr;
    if constexpr (XM) {
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)       // (clamped tile indices, no guards: guarded loads are issued and waited for one by one)
            mw[t8] = mw_row[(size_t)min(t8, ntiles - 1) * a.Ppad];
        mw_next = mw[0];
    }
    auto fetch = [&](int t) {
        const char* src = img + (size_t)t * BL::BYTES + BL::OFF_C;
        st0 = *reinterpret_cast<const v4i*>(src + tid * 16);
        if (tid + NTH < NPC) st1 = *reinterpret_cast<const v4i*>(src + (tid + NTH) * 16);
        if (tid < 8) stz = *reinterpret_cast<const f32x4*>(nz + t * 32 + tid * 4);
    };
    auto stash = [&](int b) {
        char* tile = smem + b * TILE;
        *reinterpret_cast<v4i*>(tile + 256 + tid * 16) = st0;
        if (tid + NTH < NPC) *reinterpret_cast<v4i*>(tile + 256 + (tid + NTH) * 16) = st1;
        if (tid < 8) *reinterpret_cast<f32x4*>(tile + tid * 16) = stz;
    };
    fetch(0);
    if constexpr (XM) {
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) asm volatile("" : "+v"(mw[t8]));
    }
    stash(0);
    const int crow = (h * 32 + r) * 16;
    for (int t = 0; t < ntiles; ++t) {
        uint32_t mwd = 0u;
        if constexpr (XM) {
            if (mw_pre) {
                mwd = mw[0];
#pragma unroll
                for (int t8 = 1; t8 < 8; ++t8) mwd = t == t8 ? mw[t8] : mwd;
            } else {
                mwd = mw_next;
                mw_next = mw_row[(size_t)min(t + 1, ntiles - 1) * a.Ppad];
            }
        }
        if (t + 1 < ntiles) fetch(t + 1);                 // in flight during this tile's arithmetic
        // tile t is in buffer t & 1; everybody is done with the other one.  (An LDS-only barrier: __syncthreads() is also a fence of
        // global memory - hipcc puts s_waitcnt vmcnt(0) in front of it, i.e. it waited for the loads just issued, and every tile of
        // the block cost a whole memory latency: found in round 5, dg_prep.hip k_cd_mask3 had the same)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char* tile = smem + (t & 1) * TILE;
        if (act) {
            f32x16 yc = f32x16{};
#pragma unroll
            for (int k = 0; k < NKD; ++k) {
                const f16x8 af = *reinterpret_cast<const f16x8*>(tile + 256 + crow + k * 1024);
                yc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(f16x8, Rc[k]), yc, 0, 0, 0);
            }
            f16x8 g8[2];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(tile + (8 * i4 + 4 * h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * i4 + e;
                    const float fdv = fmaf(nz_lane, v4[e], c0);
                    const bool on = XM ? ((mwd >> ((i & 3) + 8 * (i >> 2) + 4 * h)) & 1u) != 0u : (yc[i] >= a.dep_lo && yc[i] <= a.dep_hi);
                    const float g = on ? fdv : 0.f;      // d clamp(cd) / d cd
                    lsum = fmaf(fdv, fminf(fmaxf(yc[i], a.dep_lo), a.dep_hi), lsum);           // clamp(cd) (dd - shift)
                    g8[i >> 3][i & 7] = (_Float16)g;
                }
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int d = 0; d < NDF; ++d) {
                    const f16x8 bf = *reinterpret_cast<const f16x8*>(tile + 256 + (BL::OFF_P - BL::OFF_C) + (h * KD + 32 * d + r) * 16 + sp * (2 * KD * 16));
                    acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g8[sp], bf, acc[d], 0, 0, 0);
                }
        }
        if (t + 1 < ntiles) stash((t + 1) & 1);           // (its readers of tile t-1 are behind this iteration's barrier)
    }
    // raw gradient tile (accumulator order, as the fused kernel's) and the block's loss partial sum
    if (act && a.dep_dR) {
        float* base = a.dep_dR + ((size_t)n * ntiles + rtile) * (32 * KD) + lane * 4;
#pragma unroll
        for (int d = 0; d < NDF; ++d)
            if (32 * d + r < a.D) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 o = {acc[d][4 * g], acc[d][4 * g + 1], acc[d][4 * g + 2], acc[d][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(base + (d * 4 + g) * 256) = o;
                }
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
    __syncthreads();                                       // the tile buffers are free: reuse their head for the reduction
    float* red = reinterpret_cast<float*>(smem);
    int* flag = reinterpret_cast<int*>(smem + 64);
    if (lane == 0) red[wid] = act ? lsum : 0.f;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w <= GS_CW; ++w) s += red[w];
        a.dep_part[(size_t)dbid * 2] = s;
        a.dep_part[(size_t)dbid * 2 + 1] = 0.f;
        __threadfence();                                   // this block's outputs are visible before its ticket is
        *flag = atomicAdd(a.dep_ticket, 1u) == (unsigned)a.dep_blocks - 1u;
    }
    __syncthreads();
    if (*flag && wid == 0 && a.fin.out) {                  // the last depth block: every partial sum of the call is complete
        __threadfence();
        finish_scalars(a.fin, lane);
    }
}

template <int NKF, int NKD, bool XM, bool HALF = false>      // HALF: the output as DgScatterSrc.half tiles (fp16, projected, not yet divided by ||c||)
__device__ __forceinline__ void gs_body(const DgGsArgs& a, const uint32_t* dep_maskbits) {
    using BL = BlobT<NKF, NKD>;
    constexpr int KD = BL::KD, NDF = KD / 32, TS = GS_TS;
    constexpr int PB = BL::BYTES - BL::OFF_P, CH = PB / 1024;     // bytes / DMA instructions of a P part
    static_assert(GS_NB - 2 <= 4 && 4 * CH < 64, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];   // [GS_NB][PB] ring, [GS_CW][32*TS] scratch
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    // the fused kernel wrote the G tiles image-major in ascending image order: walk the images in descending order (and all
    // pair-sets of an image together) so that the most recently written tiles are read first, while the Infinity Cache has them
    // XCD-aware order (blocks are dealt round-robin over the 8 XCDs): the blocks of one image - they all read the same R
    // operand - get consecutive logical ids on one XCD, so its P parts are fetched into one L2 once
    // the depth term's blocks FIRST (they are long latency chains: started last they would be this launch's tail), then the
    // G-stream blocks
    if ((int)blockIdx.x < a.dep_blocks) {
        if (DG_DBG(a.debug) & 8192) return;                 // (ablation: no depth-term blocks)
        gs_depth_block<NKF, NKD, XM>(a, dep_maskbits, (int)blockIdx.x, gs_smem);
        return;
    }
    if (DG_DBG(a.debug) & 16384) return;                    // (ablation: the depth-term blocks alone)
    int bid;
    {
        const int nwg = (int)gridDim.x - a.dep_blocks, orig = (int)blockIdx.x - a.dep_blocks;
        const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    }
    const int nbx = (a.Ppad / 32 + GS_CW - 1) / GS_CW;
    const int bx = bid % nbx, by = (bid / nbx) % a.njobs, bz = bid / (nbx * a.njobs);
    const int n = a.B - 1 - bz;
    const DgGsJob& J = a.jobs[by];
    const int ntS = a.Ppad >> 5;
    const int gdbg = DG_DBG(a.debug);
    const int nt = (gdbg & 4096) ? 1 : ntS;              // (ablation: one R tile only)
    const int nR = J.ridx ? (int)J.ridx[n] : n;
    if (wid == GS_CW) {
        // ---- producer: P parts of R tiles 0..nt-1 through the ring
        const char* Pg = J.Rop + (size_t)nR * ntS * BL::BYTES + BL::OFF_P + lane * 16;
        const uint32_t ring_a = lds_addr(gs_smem);
        auto issue = [&](int t) {
            const uint32_t dst = ring_a + (t % GS_NB) * PB;
            const char* src = Pg + (size_t)t * BL::BYTES;
#pragma unroll
            for (int c = 0; c < CH; ++c) dma16(src + c * 1024, dst + c * 1024);
        };
        for (int t = 0; t < GS_NB - 1 && t < nt; ++t) issue(t);
        for (int rt = 0; rt < nt; ++rt) {
            gs_wait_tiles<CH>(min(GS_NB - 2, nt - 1 - rt));      // tile rt has landed
            __builtin_amdgcn_s_barrier();                          // ... and the consumers are done with tile rt-1
            if (rt + GS_NB - 1 < nt) issue(rt + GS_NB - 1);        // into the buffer tile rt-1 occupied
        }
        return;
    }
    const int st = bx * GS_CW + wid;                        // S tile of this wave
    // k_corr_main's partial sums -> output scalars: by a consumer wave that has no S tile (after its barriers, beside the
    // other waves' work), or by wave 0 of the first block when every consumer is busy
    const bool spare = (a.Ppad / 32) % GS_CW != 0;
    if (st >= ntS) {                                        // nothing to do but keep the barrier count
        for (int rt = 0; rt < nt; ++rt) __builtin_amdgcn_s_barrier();
        if (a.fin.out && a.dep_blocks == 0 && bid == nbx - 1 && wid == GS_CW - 1) finish_scalars(a.fin, lane);
        return;
    }
    if (a.fin.out && a.dep_blocks == 0 && !spare && bid == 0 && wid == 0) finish_scalars(a.fin, lane);
    const int nS = J.sidx ? (int)J.sidx[n] : n;
    f32x16 acc[NDF];
#pragma unroll
    for (int f = 0; f < NDF; ++f) acc[f] = f32x16{};
    char* T = gs_smem + GS_NB * PB + wid * (32 * TS);
    // ds_read_b64_tr_b16 addressing: lane l of a 16-lane group supplies row (l>>2)&3 and chunk 4*((l>>4)&1) + (l&3) of the block
    const int tr_a = (lane >> 2) & 3, tr_c = 4 * ((lane >> 4) & 1) + (lane & 3);
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const v4i* Gbase = reinterpret_cast<const v4i*>(J.G) + ((size_t)n * ntS + st) * ntS * 128 + lane;      // [image][S tile][R tile]: one sequential 2 KiB-per-step stream per wave
    const size_t gstride = 128;                             // v4i per R tile step
    auto load_g = [&](int rt, v4i (&g)[2]) {
        const v4i* gp = Gbase + (size_t)((gdbg & 64) ? 0 : rt) * gstride;
        g[0] = __builtin_nontemporal_load(gp);
        g[1] = __builtin_nontemporal_load(gp + 64);
    };
    v4i gring[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < nt) load_g(k, gring[k]);
    for (int rt0 = 0; rt0 < nt; rt0 += 3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int rt = rt0 + k;
            if (rt < nt) {
                // own scratch: no cross-wave hazard, so the transposition writes may start before the barrier
                const v4i g0 = gring[k][0], g1 = gring[k][1];
                const uint32_t w[8] = {(uint32_t)g0[0], (uint32_t)g0[1], (uint32_t)g0[2], (uint32_t)g0[3],
                                       (uint32_t)g1[0], (uint32_t)g1[1], (uint32_t)g1[2], (uint32_t)g1[3]};
                // [p][q] image, 64-byte rows of 8 chunks (4 q each), chunk index XOR-ed with p&7: the lane (p = r, h) owns
                // the chunks 2g + h of its row: elements 4g..4g+3 = q 8g+4h .. +3, packed in w[2g], w[2g+1]
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<uint2*>(T + r * 64 + (((2 * g + h) ^ (r & 7)) * 8)) = make_uint2(w[2 * g], w[2 * g + 1]);
                if (rt + 3 < nt) load_g(rt + 3, gring[k]);
                __builtin_amdgcn_s_barrier();               // P part of tile rt is in the ring
                asm volatile("" ::: "memory");
                if (!(gdbg & 32)) {
                    const char* P = gs_smem + (rt % GS_NB) * PB;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        // A fragment of dS = G^T x: element e of lane (q, h) is G[q][p], p = 16 ks + 8 (e>>2) + 4 h + (e&3) (the
                        // position order of the P part's granule 2 ks + h); two transposing reads of 4 image rows each
                        f16x8 afrag;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int prow = 16 * ks + 8 * u + 4 * h + tr_a;
                            typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
                            const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (lds_s16x4_p)(T + prow * 64 + ((tr_c ^ (prow & 7)) * 8)));
                            const f16x4 tf = __builtin_bit_cast(f16x4, t);
#pragma unroll
                            for (int e = 0; e < 4; ++e) afrag[4 * u + e] = tf[e];
                        }
#pragma unroll
                        for (int f = 0; f < NDF; ++f) {
                            const f16x8 bfrag = *reinterpret_cast<const f16x8*>(P + ((2 * ks + h) * KD + 32 * f + r) * 16);
                            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, bfrag, acc[f], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if (gdbg & 2048) { if (acc[0][0] == 1.2345f) J.dS[0] = acc[1][0] + acc[2][0]; return; }   // (ablation: no epilogue)
    // normalisation backward: acc[f][i] is (q = (i&3)+8*(i>>2)+4*h, channel 32 f + r);  dc = (dx - x <x,dx>) / ||c||
    const char* Cp = J.Sop + ((size_t)nS * ntS + st) * BL::BYTES + BL::OFF_C;
    static_assert(32 * GS_TS >= DG_XROWS_LDS, "scratch too small for the code rows");
    _Float16 x[NDF][16];
    dg_load_code_rows<NDF>(Cp, T, lane, x);
    float dot[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dot[i] = 0.f;
#pragma unroll
    for (int f = 0; f < NDF; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) dot[i] = fmaf((float)x[f][i], acc[f][i], dot[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) dot[i] = half_sum(dot[i]);
    if constexpr (HALF) {
        // fp16 tiles: (dx - x <x, dx>) - the division by ||c|| is the consumer's (k_combine_out), so that the stored value is bounded
        // like dx itself whatever the norm of a code vector - in pieces of eight accumulator elements, [2][64][8] per channel group
        _Float16* outh = reinterpret_cast<_Float16*>(J.dS) + ((size_t)n * ntS + st) * (32 * KD) + lane * 8;
#pragma unroll
        for (int f = 0; f < NDF; ++f)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                f16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = 8 * sp + e;
                    const int pos = st * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    v[e] = (_Float16)(pos < a.P ? (acc[f][i] - (float)x[f][i] * dot[i]) : 0.f);
                }
                if (32 * f + r < a.D) *reinterpret_cast<f16x8*>(outh + f * 1024 + sp * 512) = v;
            }
        return;
    }
    float inv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int pos = st * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        inv[i] = pos < a.P ? J.ScInv[(size_t)nS * a.Ppad + pos] : 0.f;
    }
    float* out = J.dS + ((size_t)n * ntS + st) * (32 * KD) + lane * 4;
#pragma unroll
    for (int f = 0; f < NDF; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * g + e;
                v[e] = (acc[f][i] - (float)x[f][i] * dot[i]) * inv[i];
            }
            if (32 * f + r < a.D) *reinterpret_cast<f32x4*>(out + (f * 4 + g) * 256) = v;      // padding channels are never read
        }
}

template <int NKF, int NKD, bool HALF = false>
__global__ __launch_bounds__((GS_CW + 1) * 64, 4) void k_gs(const DgGsArgs a) { gs_body<NKF, NKD, false, HALF>(a, nullptr); }
// the same launch with the exact clamp masks of the intra pair-set in the depth term's blocks (small sample grids: k_cd_mask;
// 256 registers per wave - see gs_depth_block).  Its own kernel and argument list: the plain one is at a register cliff where an
// eight-byte longer argument block alone cost 9 us at the headline (13 instead of 9 spilled registers).
template <int NKF, int NKD, bool HALF = false>
__global__ __launch_bounds__((GS_CW + 1) * 64, 2) void k_gs_xm(const DgGsArgs a, const uint32_t* dep_maskbits) {
    gs_body<NKF, NKD, true, HALF>(a, dep_maskbits);
}

// depth_only: the depth term's blocks alone (DG_EXACT_MASKS on the dense grid: the G-stream blocks run as a launch of the plain
// kernel with two blocks per CU, the depth blocks - which need the intra pair-set's mask words - as one of the 256-register form)
hipError_t dg_launch_gs(const DgGsArgs& a, const uint32_t* dep_maskbits, hipStream_t stream, bool depth_only, bool half_out) {
    dim3 grid((depth_only ? 0 : ((a.Ppad / 32 + GS_CW - 1) / GS_CW) * a.njobs * a.B) + a.dep_blocks), block((GS_CW + 1) * 64);
    if (grid.x == 0) return hipSuccess;
    DgGsArgs a2 = a;
#ifdef DG_DEVTOOLS
    if (const char* dbg = getenv("DG_DEBUG")) a2.debug = atoi(dbg);   // developer ablation switches (timing only)
#endif
    const int smem = GS_NB * 4 * a.KD * 16 + GS_CW * 32 * GS_TS;
    if (half_out) {          // fp16 output tiles: the widths of k_corr2 only (the identity grid's plan asks for them there)
        if (a.KF != 384 || a.KD != 96) return hipErrorInvalidValue;
        const void* kern = dep_maskbits ? reinterpret_cast<const void*>(k_gs_xm<24, 6, true>) : reinterpret_cast<const void*>(k_gs<24, 6, true>);
        hipError_t e = dg_set_max_smem(kern, smem);
        if (e != hipSuccess) return e;
        if (dep_maskbits) hipLaunchKernelGGL((k_gs_xm<24, 6, true>), grid, block, smem, stream, a2, dep_maskbits);
        else hipLaunchKernelGGL((k_gs<24, 6, true>), grid, block, smem, stream, a2);
        return hipGetLastError();
    }
#define DG_GS(NKF_, NKD_)                                                                                               \
    if (a.KF == NKF_ * 16 && a.KD == NKD_ * 16) {                                                                        \
        const void* kern = dep_maskbits ? reinterpret_cast<const void*>(k_gs_xm<NKF_, NKD_>)                             \
                                        : reinterpret_cast<const void*>(k_gs<NKF_, NKD_>);                               \
        hipError_t e = dg_set_max_smem(kern, smem);                                                                      \
        if (e != hipSuccess) return e;                                                                                   \
        if (dep_maskbits) hipLaunchKernelGGL((k_gs_xm<NKF_, NKD_>), grid, block, smem, stream, a2, dep_maskbits);        \
        else hipLaunchKernelGGL((k_gs<NKF_, NKD_>), grid, block, smem, stream, a2);                                      \
        return hipGetLastError();                                                                                        \
    }
    DG_GS(8, 6) DG_GS(8, 8) DG_GS(24, 6) DG_GS(24, 8) DG_GS(48, 6) DG_GS(48, 8)
#undef DG_GS
    return hipErrorInvalidValue;
}

// ---- launch helpers (host) ------------------------------------------------------------------
template <int NKF, int NKD, int NWAVES, int RF, bool GRAD, bool MAT, bool SIMPLE, int NKC = NKD>
static hipError_t launch_corr_t(const DgCorrArgs& args, hipStream_t stream) {
    using BL = BlobT<NKF, NKD>;
    const bool big = BL::BYTES > 48 * 1024;
    const int smem = (big ? 2 : 3) * (BL::BYTES + 256) + (big ? 0 : NWAVES * RF * (BL::OFF_P - BL::OFF_C)) + NWAVES * 2 * 4;
    auto kern = k_corr_main<NKF, NKD, NWAVES, RF, GRAD, MAT, SIMPLE, NKC>;
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
    if (e != hipSuccess) return e;
    const int grid = args.njobs * args.B * args.nrb;
    DgCorrArgs a2 = args;
    int smem2 = smem;
#ifdef DG_DEVTOOLS
    if (const char* dbg = getenv("DG_DEBUG")) a2.debug = atoi(dbg);   // developer ablation switches (timing only, results invalid)
    const char* stamp_file = getenv("DG_STAMPS");                     // developer aid: phase time stamps of one block
    static uint32_t* stamp_buf = nullptr;
    if (stamp_file && smem + NWAVES * 400 <= 160 * 1024) {
        if (!stamp_buf && hipMalloc(&stamp_buf, NWAVES * 400) != hipSuccess) stamp_buf = nullptr;
        if (stamp_buf) {
            a2.stamps = stamp_buf; smem2 = smem + NWAVES * 400;
            (void)dg_set_max_smem(reinterpret_cast<const void*>(kern), smem2);
        }
    }
    const char* blog_file = getenv("DG_BLOCKLOG");                    // developer aid (stamp build): per-block timeline
    static unsigned long long* blog_buf = nullptr;
    if (blog_file) {
        if (!blog_buf && hipMalloc(&blog_buf, 8192 * 64) != hipSuccess) blog_buf = nullptr;
        if (blog_buf && grid <= 8192) a2.blocklog = blog_buf;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), smem2, stream, a2);
#ifdef DG_DEVTOOLS
    if (a2.blocklog) {
        static unsigned long long hostb[8192 * 8];
        if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(hostb, blog_buf, (size_t)grid * 64, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* fp = fopen(blog_file, "wb")) { fwrite(hostb, 8, (size_t)grid * 8, fp); fclose(fp); }
        }
    }
    if (a2.stamps) {
        uint32_t host[8 * 100];
        if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(host, stamp_buf, NWAVES * 400, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* fp = fopen(stamp_file, "wb")) { fwrite(host, 4, NWAVES * 100, fp); fclose(fp); }
        }
    }
#endif
    return hipGetLastError();
}

// KF in {128, 384, 768}, KD in {96, 128}; nwaves = waves per block (8: two per SIMD, 256 registers each; 4: one per
// SIMD).  mode: 0 = loss only, 1 = loss + gradients, 2 = materialise cd / loss tensors.
hipError_t dg_launch_corr(const DgCorrArgs& args, int KF, int KD, int nwaves, int mode, hipStream_t stream) {
    const bool simple = args.lo == 0.f && args.hi > 1e30f;   // zero_clamp without stabalize: clamp(cd) == cd * mask
#define DG_CASE(NKF_, NKD_, NW_)                                                                        \
    if (KF == NKF_ * 16 && KD == NKD_ * 16 && nwaves == NW_) {                                          \
        if (simple) {                                                                                   \
            if (mode == 1) return launch_corr_t<NKF_, NKD_, NW_, 1, true, false, true>(args, stream);   \
            if (mode == 2) return launch_corr_t<NKF_, NKD_, NW_, 1, false, true, true>(args, stream);   \
            return launch_corr_t<NKF_, NKD_, NW_, 1, false, false, true>(args, stream);                 \
        }                                                                                               \
        if (mode == 1) return launch_corr_t<NKF_, NKD_, NW_, 1, true, false, false>(args, stream);      \
        if (mode == 2) return launch_corr_t<NKF_, NKD_, NW_, 1, false, true, false>(args, stream);      \
        return launch_corr_t<NKF_, NKD_, NW_, 1, false, false, false>(args, stream);                    \
    }
    // ViT-S recipe (D = 70 <= 80): the sixth code k-step of the Y chain would multiply zero padding
    if (KF == 384 && KD == 96 && nwaves == 8 && mode == 1 && simple && args.D <= 80)
        return launch_corr_t<24, 6, 8, 1, true, false, true, 5>(args, stream);
    DG_CASE(8, 6, 4) DG_CASE(8, 8, 4)
    DG_CASE(24, 6, 4) DG_CASE(24, 6, 8) DG_CASE(24, 8, 4)
    DG_CASE(48, 6, 4) DG_CASE(48, 8, 4)
#undef DG_CASE
    return hipErrorInvalidValue;
}


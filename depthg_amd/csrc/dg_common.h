// Shared device/host definitions for the gfx950 DepthG correlation-loss kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/depthg_corr.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// Developer ablation switches (the `debug` bits of the argument blocks, the DG_DEBUG / DG_PREP_DEBUG / DG_STAMPS /
// DG_BLOCKLOG / DG_SCATTER_CG environment hooks) exist only in a `make EXTRA=-DDG_DEVTOOLS` build; in the production
// library the bits fold to zero at compile time and the hooks are not compiled.
#ifdef DG_DEVTOOLS
#define DG_DBG(x) (x)
#else
#define DG_DBG(x) 0
#endif

#define DG_EPS_NORM 1e-10f  // F.normalize eps, reference src/modules.py:790

// Position permutation inside each 32-position block of the P-major code operand, chosen so that
// the B fragment of the gradient product (k order = accumulator row order of a 32x32 MFMA tile,
// cdna guide section 3 "An accumulator tile as the next MFMA's operand") is one 16-byte read:
// position pl = 16*s + 8*u + 4*hh + v is stored at (2*s + hh)*8 + 4*u + v.
__host__ __device__ inline int dg_perm32(int pl) {
    int s = pl >> 4, u = (pl >> 3) & 1, hh = (pl >> 2) & 1, v = pl & 3;
    return (2 * s + hh) * 8 + 4 * u + v;
}

// ---------------------------------------------------------------------------------------------
// Operand layout in HBM ("blob" layout).  A prepared operand = the normalised sampled feats (bf16) and
// code (fp16) of one tensor pair, stored per image n and per tile of 32 positions as ONE contiguous blob
// that is byte-for-byte the LDS image the correlation kernel wants, so that staging a tile is a linear
// global->LDS DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) with no registers involved:
//   F part  [GF/IG groups][32 positions q][IG granules]  bf16, see dg_f_off below
//   C part  [GD granules][32 positions]    fp16, K-major code, granule-major (conflict-free as is)
//   P part  [4 granules c][KD channels d]  fp16, P-major code: granule c of channel d holds the positions
//           with dg_perm32(pl) in [8c, 8c+8)
// granule = 16 bytes = 8 elements.  KF in {128,384,768} (GF multiple of 16), KD in {96,128}.
// F part: granule g (8 channels) of tile row q.  DG_F_IG consecutive granules of a row stay together (IG*16 bytes), the 32
// rows are interleaved at that grain: [g / IG][q][IG granules], the slot inside a row's group XORed with a few row bits so
// that the 16 lanes of one ds_read_b128 pass (16 consecutive rows, one granule) cover all 64 LDS banks.  IG = 48 granules
// would be plain row-major; small IG makes the stationary operand's fragment loads (every lane = its own row) touch few
// cache lines per instruction - they are address-coalescing bound at the start of every block of k_corr_main.
#ifndef DG_F_IG
#define DG_F_IG 4
#endif
__host__ __device__ inline int dg_f_off(int q, int g) {
    constexpr int IG = DG_F_IG;
    return ((g / IG) * 32 + q) * (IG * 16) + (((g % IG) ^ ((q / (16 / IG)) % IG)) * 16);
}

struct DgBlob {
    int GF, GD, KD;
    int off_c, off_p, bytes;
    __host__ __device__ DgBlob(int KF, int KD_) : GF(KF / 8), GD(KD_ / 8), KD(KD_) {
        off_c = 32 * GF * 16;
        off_p = off_c + GD * 32 * 16;
        bytes = off_p + 4 * KD * 16;
    }
    __host__ __device__ int f(int q, int g) const { return dg_f_off(q, g); }
    __host__ __device__ int c(int q, int g) const { return off_c + (g * 32 + q) * 16; }
    __host__ __device__ int p(int d, int cc) const { return off_p + (cc * KD + d) * 16; }
};

template <int NKF, int NKD>
struct BlobT {
    static constexpr int KF = NKF * 16, KD = NKD * 16, GF = KF / 8, GD = KD / 8;
    static constexpr int OFF_C = 32 * GF * 16;
    static constexpr int OFF_P = OFF_C + GD * 32 * 16;
    static constexpr int BYTES = OFF_P + 4 * KD * 16;
    static constexpr int CHUNKS = BYTES / 1024;          // 1 KiB DMA pieces
    static constexpr int CHUNK_C0 = OFF_C / 1024;        // first chunk of the C part
    static constexpr int CHUNK_P0 = OFF_P / 1024;
    static_assert(BYTES % 1024 == 0 && OFF_C % 1024 == 0 && OFF_P % 1024 == 0, "blob parts must be KiB multiples");
    static_assert(GF % 16 == 0, "swizzle needs 16-granule groups");
};

// Gradient buffers (w.r.t. sampled code rows) are kept in MFMA accumulator order ("gradient tiles"):
//   [image][tile of 32 positions][channel group f = d/32][g = q/8][lane = (d%32) + 32*((q/4)%2)][e = q%4]   fp32
// i.e. exactly the registers of a 32x32 accumulator tile (rows = positions, lanes = channels), so that the kernels that
// produce them (k_corr_main, k_gs) and the combine kernel move 16 bytes per lane, 1 KiB per wave instruction.
// Float index of (position p, channel d) inside one image:
__host__ __device__ inline size_t dg_gtile_off(int p, int d, int DP) {
    const int q = p & 31;
    return ((size_t)(p >> 5) * (DP >> 5) + (d >> 5)) * 1024 + (q >> 3) * 256 + ((d & 31) + 32 * ((q >> 2) & 1)) * 4 + (q & 3);
}

#ifdef __HIPCC__
// Code rows of one tile for the normalisation backward, in accumulator order: x[f][i] = normalised code of position
// q = (i&3) + 8 (i>>2) + 4 (lane>>5), channel 32 f + (lane&31).  `Cp` = C part of the tile's blob (K-major granules),
// read with coalesced 16-byte loads and turned around through a per-wave LDS scratch of DG_XROWS_LDS bytes (granule rows
// padded by one granule against bank conflicts).
#define DG_XROWS_LDS (4 * 33 * 16)
template <int NDF>
__device__ __forceinline__ void dg_load_code_rows(const char* Cp, char* T, int lane, _Float16 (&x)[NDF][16]) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const int r = lane & 31, h = lane >> 5;
    v4i_ raw[NDF][2];
#pragma unroll
    for (int f = 0; f < NDF; ++f)
#pragma unroll
        for (int k = 0; k < 2; ++k) raw[f][k] = *reinterpret_cast<const v4i_*>(Cp + f * 2048 + k * 1024 + lane * 16);
#pragma unroll
    for (int f = 0; f < NDF; ++f) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int L = k * 64 + lane;
            *reinterpret_cast<v4i_*>(T + ((L >> 5) * 33 + (L & 31)) * 16) = raw[f][k];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = (i & 3) + 8 * (i >> 2) + 4 * h;
            x[f][i] = *reinterpret_cast<const _Float16*>(T + ((r >> 3) * 33 + q) * 16 + (r & 7) * 2);
        }
    }
}
#endif

#ifdef __HIPCC__
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS byte address (wave-uniform) of a pointer into the dynamic shared segment
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)p);
}

// LDS-DMA: every lane gives its own global source address; the wave writes 64 x 16 (or 64 x 4) contiguous
// bytes at the wave-uniform LDS address.  Issued through inline asm so that hipcc neither drains it with
// vmcnt(0) before unrelated LDS reads nor counts it; completion is enforced by the explicit counted
// "s_waitcnt vmcnt" + s_barrier at the top of the tile loop (cdna guide 5.7: M0 written in the same statement).
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// ... from a wave-uniform base + a 32-bit byte offset per lane (one VGPR instead of an address pair; M0 is NOT restored: for kernels
// in which nothing else reads it)
__device__ __forceinline__ const void* dg_uniform_ptr(const void* p) {      // a wave-uniform pointer hipcc holds in VGPRs -> an SGPR pair
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void dma16_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#endif

#ifdef __HIPCC__
// a * b rounded to float32 ON ITS OWN: the empty asm hides the product from the contraction pass, so a following addition
// cannot turn the pair into one fused multiply-add (needed where a CPU reference rounds twice)
__device__ __forceinline__ float dg_mul_rn(float a, float b) {
    float r = a * b;
    asm volatile("" : "+v"(r));
    return r;
}

// Sum over the 32 lanes of each half-wave (lanes 0-31 and 32-63 separately), result in every lane of the half.  DPP row
// operations instead of the LDS crossbar: inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), row totals carried
// into the odd rows (row_bcast15), then lanes 31 / 63 are broadcast.
__device__ __forceinline__ float half_sum(float v) {
#define DG_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
    DG_DPP_ADD(0x111, 0xf); DG_DPP_ADD(0x112, 0xf); DG_DPP_ADD(0x114, 0xf); DG_DPP_ADD(0x118, 0xf);
    DG_DPP_ADD(0x142, 0xa);
#undef DG_DPP_ADD
    const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return (threadIdx.x & 32) ? s1 : s0;
}
#endif

#ifdef __HIPCC__
// dg_prof_main_span: one thread per workgroup stamps the launch's span (constant 100-MHz clock) and adds its own lifetime in
// shader cycles (s_memtime) and in wall ticks to two running sums: sum of cycles / sum of ticks = the clock the CUs HELD while
// they ran the kernel (span[2] / span[3] x 0.1 GHz).  `keep` = two 64-bit words of the workgroup's LDS (the entry stamps wait
// there: no register lives across the kernel for them).
__device__ __forceinline__ void dg_span_enter(unsigned long long* span, unsigned long long* keep) {
    if (span) {
        const unsigned long long w = (unsigned long long)wall_clock64();
        unsigned long long c;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) :: "memory");
        keep[0] = w; keep[1] = c;
        __hip_atomic_fetch_min(&span[0], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void dg_span_exit(unsigned long long* span, const unsigned long long* keep) {
    if (span) {
        const unsigned long long w = (unsigned long long)wall_clock64();
        unsigned long long c;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) :: "memory");
        __hip_atomic_fetch_max(&span[1], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&span[2], c - keep[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&span[3], w - keep[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#endif

// job kinds of the fused correlation kernel
enum { DG_JOB_HELPER = 0, DG_JOB_DEPTH = 1 };

// One pass of the row-stationary correlation kernel over one pair-set.
// "R" = stationary operand (its positions live on MFMA lanes / output rows of the gradient),
// "S" = streamed operand (tiles of 32 positions through LDS).
struct DgJob {
    const char* Rop;      // operand blobs [B][Ppad/32][blob bytes] of the stationary operand
    const char* Sop;      // ... of the streamed operand
    const float* rvec;    // fp32 [B][Ppad] row means a_p . bbar (indexed by operand-1 position) or null
    const float* rimg;    // fp32 [B] per-image sums of rvec: m0 = sum / (B*P) = old_mean of the reference (modules.py:1237) or null
    const uint32_t* maskbits;   // [B][Ppad/32 (S tile)][Ppad (R position)]: bit i = 1[cd >= 0] of (S position 32 tile + i, R position), from
                                // the fp32 sampled code rows (k_cd_mask; small sample grids) - or null: the sign of the fp16-operand cd
    const float* nzR;     // fp32 [B][Ppad] depth indicators (DG_JOB_DEPTH)
    const float* nzS;
    const float* RcInv;   // fp32 [B][Ppad] 1/max(||c||,eps) of the R code operand (normalisation backward)
    const float* Scsum;   // fp32 [B][KD] column sums (over positions) of the S operand's normalised code, or null
    const int64_t* ridx;  // batch index map of R operands (null = identity)
    const int64_t* sidx;  // batch index map of S operands (null = identity)
    float* dR;            // gradient tiles (dg_gtile_off): d/d(normalised R code), unit upstream, normalisation backward pending; or null
    float* part;          // fp32 [blocks of this job][2] partial sums (sum clamp(cd)*(fd-shift), sum cd); or null
    float* out_cd;        // fp32 [B][P][P] (op1 position major) or null    (materialise; needs center_on_lane == 0)
    float* out_loss;      // fp32 [B][P][P] or null
    uint16_t* Gout;       // fp16 [B][S tile][R tile][2 k-steps][64 lanes][8] = G tiles, accumulator registers 8s..8s+7 of every lane
                          // (one contiguous KiB per k-step and wave instruction; input of k_gs) or null
    float shift;
    int32_t kind;
    int32_t center_on_lane;  // 1: R is operand 1 (rvec / nzR indexed by lane); 0: R is operand 2 (rvec by tile row)
    int32_t slot_loss;       // output scalar the loss sum of this job adds to (DG_OUT_*; -1 none)   } copied into
    int32_t slot_cd;         // ... the cd sum                                                        } DgFinishArgs
    float fin_scale;         // 1/numel of the tensor the job contributes to                         } by the host
    int32_t fold;            // k_corr2 FOLD: the intra pair-set's streamed-side gradient is formed in the fused kernel (no G tiles read)
};

#define DG_MAX_JOBS 12      // pair-sets (<= DG_MAX_NEG + 2) + the depth job
#define DG_GR_CAP 24        // k_corr2's grouped ragged blocks: listed consumers per (key, streamed image); the rest run as one-(pair-set, image) blocks

struct DgCorrArgs {
    DgJob jobs[DG_MAX_JOBS];
    int32_t njobs;
    int32_t B, P, Ppad;
    int32_t nrb;          // row blocks per image = ceil(Ppad / (waves per block * 32))
    int32_t D;            // real code channels
    float lo, hi;         // clamp bounds
    float inv_BP;         // 1 / (B*P)
    const char* dummy;    // any valid device address (source of DMA lanes that carry nothing)
    int32_t pos_w;        // > 0: positions are pixel indices y*w + x of a w x w identity grid (DG_IDENTITY_GRID); the un-reduced outputs
                          //      (materialise) are written at the reference's position x*w + y
    int32_t debug;        // developer ablation bits (0 in production)
    uint32_t* wctr;       // k_corr2's persistent workgroups: [0..7] items handed out so far per XCD (beyond each workgroup's first), [8]
                          // workgroups that have left; all zero at launch (k_colmean) and again when the last workgroup leaves; null: static walk
    unsigned long long* span;   // measurement aid (dg_prof_main_span): [0] min of the workgroups' entry times, [1] max of their exit times, [2] / [3] sums of their lifetimes in shader cycles / wall ticks; or null
    int32_t half_tiles;   // k_corr2: 1 = the raw gradient tiles (DgJob.dR) are written as fp16 (DgScatterSrc.half)
    uint32_t* stamps;     // developer timing stamps (null in production)
    unsigned long long* blocklog;   // developer block timeline: [block][8] = hw id, xcc id, 4 wall-clock stamps (null in production)
    // ragged last row blocks grouped by streamed operand (dg_corr2.hip; lists written by k_group_ragged); gr_list null: off
    const int32_t* gr_list;    // [nkeys][B][DG_GR_CAP]: pair-set | image << 8 of the consumers of (key, streamed image)
    const int32_t* gr_count;   // [nkeys][B]: how many of them are listed
    const int16_t* gr_rank;    // [helper jobs][B]: rank of (pair-set, image) among the consumers of its (key, streamed image)
    int32_t gr_nkeys, gr_cpb;  // keys; consumers per grouped block = 8 / (row tiles of the ragged row block)
    int32_t gr_blocks_per_image;             // sum of gr_nblk over the keys
    int8_t gr_key[DG_MAX_JOBS];              // key of helper job j (same streamed operand array = same key)
    int8_t gr_first[DG_MAX_JOBS];            // first pair-set of key k
    int32_t gr_nblk[DG_MAX_JOBS];            // grouped blocks per streamed image of key k
};

// Final reduction of the per-block partial sums of k_corr_main into the output scalars.  It runs in the NEXT launch on the
// stream (the first block of k_gs on a gradient pass, the one-wave k_finish otherwise), so the fused kernel needs neither
// atomics nor fences for it.
struct DgFinishArgs {
    const float* part[DG_MAX_JOBS];   // per job: [nblk][2] partial (loss, cd) sums; null: job contributes nothing
    int32_t slot_loss[DG_MAX_JOBS];   // output scalar the loss sum of job j adds to (DG_OUT_*; -1 none)
    int32_t slot_cd[DG_MAX_JOBS];
    float scale[DG_MAX_JOBS];         // 1/numel of the tensor the job contributes to
    int32_t njobs, nblk, B, P;
    int32_t nblk_job[DG_MAX_JOBS];    // partial sums of job j if not nblk (0: nblk)
    const float* nzsum;               // [B] per-image sums of the depth indicators (mean(dd)) or null
    float* out;                       // [DG_OUT_COUNT]; null: nothing to do
    float wtot[4];                    // weights of the four loss means in out[DG_OUT_TOTAL]
};

// ---- argument blocks of the helper kernels (one definition shared by kernels and host API)

struct DgTransposeArgs {    // NCHW (B,K,h,w) fp32 -> NHWC (B,h*w,K4) fp32 for up to four maps in one launch
    const float* src[4];
    float* dst[4];
    int32_t K[4], K4[4], HW[4];      // per map: channels, padded channels, pixels (the code maps may differ in size from the feature maps)
    int32_t nmaps;
};

struct DgGatherJob {
    const float* src;        // NHWC fp32 [B][h*w][K4]
    const float* coords;     // [B][S][S][2]
    const int64_t* srcidx;   // batch map (image n is read from src[srcidx[n]]) or null
    char* blob;              // operand blobs [B][Ppad/32][blob bytes]
    float* inv_norm;         // [B][Ppad] or null (code)
    float* colpart;          // [B][Ppad/32][Kpad] per-tile column sums of the normalised rows or null (feats)
    int32_t K, K4, Kpad;
    int32_t is_code;         // 1: fp16 code (C and P parts), 0: bf16 feats (F part)
    int32_t h, w;            // size of the map `src` holds (not read in direct mode)
    const float* ext_inv;    // [B][P] or null: 1 / norm of the sampled vector over ALL channels, of which this job holds a chunk (dg_corr_forward_extnorm)
};
#define DG_MAX_GATHER 20
// the jobs of a call with general coordinates that depend on nothing but its inputs (k_pre_general; dg_post.hip)
struct DgPreArgs {
    uint64_t seed; unsigned long long* state; int64_t* perms; int32_t count;      // draws (count == 0: none)
    const float* depth; float* nz; float* nzsum; int32_t dH, dW;                  // depth indicators
    const float* coords1; const float* coords2; char* taps;                       // inverse tap records [2][B]
    int32_t B, h, w, S, Sh, P, Ppad;
    unsigned int* zero_word;      // a word this launch sets to 0 (the ticket of the fused small-grid kernel), or null
};

// exact clamp masks of the small sample grids (k_cd_mask; dg_prep.hip)
struct DgCdMaskArgs {
    const float* rowsR;                      // sampled code rows of operand 1: (B, P, D4) fp32
    const float* rowsS[DG_MAX_NEG + 2];      // ... of the streamed operand of pair-set t
    const int64_t* sidx[DG_MAX_NEG + 2];     // batch map of the streamed operand (null: identity)
    uint32_t* bits[DG_MAX_NEG + 2];          // [B][Ppad/32][Ppad] out
    int32_t T, B, P, Ppad, D, D4;
};

struct DgGatherArgs {
    DgGatherJob jobs[DG_MAX_GATHER];
    int32_t njobs, B, S, Sh, P, Ppad, KF, KD;   // sample grid: Sh rows x S columns (Sh == S, or 1 with DG_LINE_GRID)
    int32_t direct;          // 1: src holds the SAMPLED rows already, [B][P][K4] per job (k_plane_sample): no taps, no batch map
    // cd.T > 0: the exact clamp masks ride in this launch (they read the sampled code rows, like the gather: one launch less on the
    // small sample grids) - blockIdx.z >= njobs: slice (z - njobs) / cd_xper is pair-set t, the rest extends blockIdx.x
    DgCdMaskArgs cd;
    int32_t cd_xper;
    // pre_blocks > 0: the depth indicators and the inverse tap records of the sample() adjoint (the roles of k_pre_general that nothing
    // in front of the fused kernel reads) ride here too, in the LAST z slices: block id (linear over the extra slices) < pre_nz: the
    // depth indicators of image id; then 2 B tap-record blocks.  (pre.count is not used: the draw of the batch maps keeps its launch)
    DgPreArgs pre;
    int32_t pre_blocks, pre_nz, pre_z0;
};

struct DgPlaneArgs {        // k_plane_sample: sample() of all operands straight from the NCHW maps, small sample grids
    const float* src[4];     // orig_feats, orig_feats_pos, orig_code, orig_code_pos  (B,K,h,w) fp32
    int32_t K[4], K4[4];
    float* rows[DG_MAX_NEG + 2][2];   // [operand][0 feats, 1 code]: sampled rows (B, P, K4) fp32, channels K..K4-1 zero
    const float* coords1;
    const float* coords2;
    const int64_t* perms;    // [nops - 2][B] batch maps of the negatives (operand o >= 2 of image n reads image perms[o-2][n])
    int32_t nops, B, h, w, S, Sh, P;
    int32_t tap_consumers;   // (set by the launcher) consumers whose tap table is held in LDS together
    int32_t feats_bf16;      // 1: rows[.][0] are bf16 (B, P, K4) - the fused small-grid kernel's input (K4 then a multiple of 8)
};

#ifdef __HIPCC__
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): counter (ctr, 0, 0, 0), key = the 64-bit seed
__device__ __forceinline__ uint32_t dg_philox(uint64_t seed, uint32_t ctr) {
    uint32_t c0 = ctr, c1 = 0u, c2 = 0u, c3 = 0u, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// keys == nullptr: the uniform keys are drawn here (Philox, counter = row * B + i), one launch instead of rand + sort.
// state != nullptr: the Philox key is {state[0] (seed), state[1] (draws so far)} READ FROM THE DEVICE, and the block that
// finishes last advances state[1] (ticket in state[2]) - a launch recorded in a hipGraph then draws fresh permutations on
// every replay, which a seed passed by value cannot.

// One row of super_perm (src/modules.py:1184-1188): rank of every key inside the row (ties by index) = position of that index in
// the argsort, then the fixed-point bump modulo B.  keys: given, or drawn here from (seed | state).  `state` = {seed, draws so
// far, ticket}: the last of the `nrows` rows to finish advances the draw count (device-resident generator: hipGraph-safe).
// Called by the whole block (256 threads), sk = B floats of LDS.
__device__ __forceinline__ void dg_super_perm_row(const float* __restrict__ keys, uint64_t seed, unsigned long long* __restrict__ state,
                                                  int B, int64_t* __restrict__ out, int row, int nrows, float* sk) {
    const float* kr = keys ? keys + (size_t)row * B : nullptr;
    unsigned long long draw = 0;
    if (state) { seed = state[0]; draw = state[1]; }
    const uint64_t key = seed + 0x9E3779B97F4A7C15ull * draw;
    for (int i = threadIdx.x; i < B; i += 256)
        sk[i] = kr ? kr[i] : (float)(dg_philox(key, (uint32_t)(row * B + i)) >> 8) * (1.0f / 16777216.0f);
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {
        const float ki = sk[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (sk[j] < ki) || (sk[j] == ki && j < i);
        out[(size_t)row * B + rank] = (int64_t)((i == rank ? i + 1 : i) % B);
    }
    if (state && threadIdx.x == 0) {
        // every row has read the state before it takes its ticket; the last ticket advances the draw count
        __threadfence();
        if (atomicAdd(&state[2], 1ull) == (unsigned long long)nrows - 1) {
            state[1] = draw + 1;
            state[2] = 0;
            __threadfence();
        }
    }
}
#endif

#ifdef __HIPCC__
// depth (B,1,H,W) -> nz[n][p] over the S x S resize, p = i*S + j (row major)
__device__ __forceinline__ float depth_nz_at(const float* __restrict__ depth, int n, int p, int H, int W, int Sh, int S) {
    float out = 0.f;
    if (p < Sh * S) {
        const int i = p / S, j = p - i * S;
        const float sy = Sh > 1 ? (float)(H - 1) / (float)(Sh - 1) : 0.f;
        const float sx = S > 1 ? (float)(W - 1) / (float)(S - 1) : 0.f;
        // every product is rounded on its own (dg_mul_rn), as in the torch operator: contracted into the subtraction below,
        // scale * index leaves a 1e-7 weight where the rounded source coordinate is a whole pixel - enough to pull a non-zero
        // neighbour into a pixel of zero depth and flip its indicator (found by scripts/fuzz_parity.py, seed 323)
        const float fy = dg_mul_rn(sy, (float)i), fx = dg_mul_rn(sx, (float)j);
        int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
        const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* d = depth + (size_t)n * H * W;
        const float top = dg_mul_rn(d[(size_t)y0 * W + x0], lx0) + dg_mul_rn(d[(size_t)y0 * W + x1], lx1);
        const float bot = dg_mul_rn(d[(size_t)y1 * W + x0], lx0) + dg_mul_rn(d[(size_t)y1 * W + x1], lx1);
        const float v = dg_mul_rn(top, ly0) + dg_mul_rn(bot, ly1);
        out = v / fmaxf(fabsf(v), DG_EPS_NORM);
    }
    return out;
}

// all positions of image n by one block of 256 threads, plus their sum (mean(dd) = mean_n (sum_p nz)^2 / P^2)
// `pixel_order` (identity grid): position p is pixel p = y*S + x of the map, whose sample() output index is (i, j) = (x, y), i.e.
// the reference's position x*S + y - that is where the resized depth is read
__device__ __forceinline__ void depth_nz_image(const float* __restrict__ depth, float* __restrict__ nz, float* __restrict__ nzsum,
                                               int n, int H, int W, int Sh, int S, int Ppad, bool pixel_order = false) {
    __shared__ float wred[4];
    float s = 0.f;
    for (int p = threadIdx.x; p < Ppad; p += 256) {
        const int pref = (pixel_order && p < Sh * S) ? (p % S) * S + p / S : p;
        const float v = depth_nz_at(depth, n, pref, H, W, Sh, S);
        nz[(size_t)n * Ppad + p] = v;
        s += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) nzsum[n] = wred[0] + wred[1] + wred[2] + wred[3];
}
#endif

// ---- the fused small-sample-grid path (dg_small.hip; round 5): Ppad <= 160 positions per image (the S = 11 / 12 recipes of
// paper_reproduction.sh:5-14 and everything below), any feature width.  ONE launch per call replaces operand building, column / row
// means, exact clamp masks, the correlation and the streamed-side gradient: a block owns one (image, pair-set[, half of the
// stationary tiles]) and reads the SAMPLED fp32 rows of its two operands once.
struct DgSmallArgs {
    const void* rowsF[DG_MAX_NEG + 2];    // [operand][B][P][C4] sampled feature rows, bf16 (k_plane_sample / k_gather_rows), C4 a multiple of 128 (whole chunks), channels C..C4-1 zero
    const float* rowsC[DG_MAX_NEG + 2];   // [operand][B][P][D4] sampled code rows
    int32_t T, B, P, Ppad, C4, D, D4, KD; // T pair-sets; KD in {96, 128}: padded code width of the gradient tiles
    int32_t opS[DG_MAX_NEG + 2];          // streamed operand of pair-set t (t itself; with DG_SHARED_COORDS the negatives stream operand 0 ...
    const int64_t* sidx[DG_MAX_NEG + 2];  // ... of image sidx[t][n] - the batch map - instead of image n; null: image n)
    int32_t pointwise, depth, grad;
    float lo, hi;                         // clamp bounds
    float shift[DG_MAX_NEG + 2], shift_depth;
    const float* nz;                      // [B][Ppad] depth indicators (depth term)
    const float* nzsum;                   // [B]
    float* dRA[DG_MAX_NEG + 3];           // gradient tiles (dg_gtile_off), raw (normalisation backward pending): d/d(normalised operand-0 code)
                                          // of pair-set t from sum_q -G[p][q] y_q; [T] = the depth term
    float* dRA2[DG_MAX_NEG + 2];          // pointwise: the same with -G replaced by the clamp mask (factor old_mean_t, see om)
    float* dRB[DG_MAX_NEG + 2][2];        // final tiles (normalisation backward applied) of the streamed operand, one per half of the R tiles
    float* dRB2[DG_MAX_NEG + 2][2];       // pointwise: mask form
    float* part;                          // [T + 1][B][nsplit][4]: sum clamp(cd)(fd' - shift), sum clamp(cd), sum fd, sum cd
    float* om;                            // [T] out: old_mean of pair-set t (0 without pointwise) - the factor of the "2" gradient sets
    unsigned int* ticket;                 // zero at launch; the block that finishes last reduces `part` into `out`
    char* xop;                            // operand-0 blobs: the C part (normalised fp16 code rows) is written for the backward tail
    float* xinv;                          // [B][Ppad] 1 / max(||code||, eps) of operand 0
    int32_t blob_bytes, blob_off_c;
    float* out;                           // [DG_OUT_COUNT]
    float wtot[4];
    int32_t nsplit;                       // blocks per (image, pair-set): 1, or 2 when the image has 5 tiles (3 + 2 stationary tiles)
    // materialise (dg_corr_materialize): the un-reduced tensors of pair-set mat_t (-1: the depth term's dd) instead of everything above
    float* out_cd;                        // [B][P][P] or null
    float* out_loss;
    int32_t mat_t, mat;
    int32_t debug;                        // developer: 1 = block 0 prints its phase stamps (DG_SMALL_DEBUG=1)
    unsigned long long* span;             // measurement aid (dg_prof_main_span) or null
};

struct DgGatherRowsArgs {   // k_gather_rows: sample() of channel-last maps into fp32 rows (code maps of another size, maps beyond the LDS)
    const float* src[2 * (DG_MAX_NEG + 2)];      // NHWC fp32 [B][h*w][K4]
    const float* coords[2 * (DG_MAX_NEG + 2)];   // [B][S][Sh][2]
    const int64_t* srcidx[2 * (DG_MAX_NEG + 2)]; // batch map or null
    void* rows[2 * (DG_MAX_NEG + 2)];            // [B][P][Kout]: fp32, or bf16 when as_bf16 (Kout then a multiple of 8, padding zeroed)
    int32_t K4[2 * (DG_MAX_NEG + 2)], h[2 * (DG_MAX_NEG + 2)], w[2 * (DG_MAX_NEG + 2)];
    int32_t Kout[2 * (DG_MAX_NEG + 2)], as_bf16[2 * (DG_MAX_NEG + 2)];
    int32_t njobs, B, S, Sh, P;
};

struct DgDenseArgs {        // identity-grid operand preparation (k_prep_dense)
    const float* src[2];     // feats NCHW fp32 (B,K,h,w): orig_feats, orig_feats_pos
    const float* code[2];    // code NCHW fp32 (B,D,h,w): orig_code, orig_code_pos
    char* blob[2];           // operand blobs 0, 1
    float* colpart[2];       // [B][h][KF] per-source-row column sums of the normalised feats
    float* inv_norm[2];      // [B][Ppad] 1/max(||code||, eps)
    float* ccolpart[2];      // [B][Ppad/32][KD] per-tile column sums of the normalised code
    const float* depth;      // (B,1,dH,dW) or null
    float* nz;               // [B][Ppad] depth indicators
    float* nzsum;            // [B] their per-image sums
    int32_t B, K, D, KF, KD, h, w, P, Ppad, dH, dW;
    int32_t debug;           // developer ablation bits (0 in production): 1 skip feats, 2 skip code, 4 skip depth
    // draw_count > 0: that many extra blocks draw the negatives' batch maps (dg_super_perm_row) into draw_out - the step's
    // k_super_perms launch rides here (dg_corr_forward_draw)
    int64_t* draw_out;
    unsigned long long* draw_state;
    uint64_t draw_seed;
    int32_t draw_count;
    int32_t code_split;      // 1: the code role only writes inv_norm (per-pixel norms from whole channel planes); the code parts of
                             //    the blobs + ccolpart come from the k_colmean launch (DgDenseCodeArgs), csum from the k_rowmean launch
    // Dropout2d of the feature maps applied HERE instead of by their producer (dg_corr_forward_masked): fkeep[o] (B,K) keep flags
    // 1 / 0 of source o or null, the kept channels scaled by fscale = 1/(1-p) - the product the producer would have written
    const float* fkeep[2];
    float fscale;
    int32_t unit;            // DG_FEATS_UNIT: the feature rows are written as they are (unit vectors, or a channel chunk of them)
    int32_t roles;           // 0: every role; else a mask of the roles THIS launch runs - 1 feats, 2 code, 4 depth indicators, 8 the draw
                             // (the launch split in two that run on two streams: dg_api.hip, exact clamp masks on the dense grid)
};

// Code operands of the identity grid from whole channel planes (extra blocks of the k_colmean launch, after the norms of
// k_prep_dense): one block per (image, operand, group of 8 channels = one 16-byte granule of the C part)
struct DgDenseCodeArgs {
    const float* code[2];    // code NCHW fp32 (B,D,h,w)
    char* blob[2];
    const float* inv_norm[2];
    float* ccolpart[2];
    int32_t B, D, KF, KD, h, w, P, Ppad;     // B == 0: not used
    // exact clamp masks (DG_EXACT_MASKS): the part of the normalised code the fp16 C part drops, (x - fp16(x)) * 2048 as fp16, in
    // the C part's own granule layout, [image][tile][KD/8][32 positions][8]; null: not wanted
    char* clo[2];
};

// Consumer lists of the grouped ragged row blocks of k_corr2 (dg_corr2.hip): for every key (= set of pair-sets that stream the
// same operand array) and every image m of that array, the (pair-set, image) pairs whose streamed operand is image m, in
// (pair-set, image) order - the first DG_GR_CAP of them as a list, and for every (pair-set, image) its rank in that order.
// Written by extra blocks of the k_colmean launch (nothing of its own to wait for: the batch maps are inputs of the call).
struct DgGroupArgs {
    const int64_t* sidx[DG_MAX_JOBS];   // batch map of the streamed operand of helper job j (null: the image itself)
    int8_t key[DG_MAX_JOBS];
    int32_t nh, nkeys, B;               // helper jobs, keys (0: no lists), images (<= 64)
    int32_t* list;                      // [nkeys][B][DG_GR_CAP]: pair-set | image << 8
    int32_t* count;                     // [nkeys][B]
    int16_t* rank;                      // [helper jobs][B]
};
#ifdef __HIPCC__
__device__ __forceinline__ void dg_group_lists(const DgGroupArgs& g, const int m, const int key, const int lane) {
    const int B = g.B;
    // (all batch-map entries first - independent loads, one latency - then the ballots in pair-set order)
    int src[DG_MAX_JOBS];
#pragma unroll
    for (int j = 0; j < DG_MAX_JOBS; ++j) {
        const bool mine = j < g.nh && g.key[j] == key && lane < B;
        const int64_t* sidx = mine ? g.sidx[j] : nullptr;
        src[j] = mine ? (sidx ? (int)sidx[lane] : lane) : -1;
    }
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < DG_MAX_JOBS; ++j) {
        const bool hit = src[j] == m;
        const unsigned long long mask = __ballot(hit);
        if (hit) {
            const int rk = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            g.rank[j * B + lane] = (int16_t)(rk < 32767 ? rk : 32767);
            if (rk < DG_GR_CAP) g.list[(key * B + m) * DG_GR_CAP + rk] = j | (lane << 8);
        }
        cnt += __popcll(mask);
    }
    if (lane == 0) g.count[key * B + m] = cnt < DG_GR_CAP ? cnt : DG_GR_CAP;
}
#endif

struct DgColmeanArgs {      // bbar[o][n][k] = (1/P) sum_groups colpart[o][n][group][k];  csum[o][n][d] = sum_tiles ccolpart[o][n][tile][d]
    const float* colpart[DG_MAX_NEG + 2];   // feats partial column sums (null: skip)
    float* bbar[DG_MAX_NEG + 2];
    __bf16* bsplit[DG_MAX_NEG + 2];         // [B][2][KF] bbar split into bf16 hi / lo
    const float* ccolpart[DG_MAX_NEG + 2];  // code partial column sums [B][Ppad/32][KD]
    float* csum[DG_MAX_NEG + 2];            // [B][KD]
    int32_t ngroups[DG_MAX_NEG + 2];   // feats partial-sum groups per image (tiles, or source rows on the dense path)
    int32_t nops, B, P, Ppad, KF, KD;
    unsigned int* zero_word;           // a word this launch sets to 0 (the depth blocks' ticket of the k_gs launch), or null
    unsigned int* zero_words9;         // nine words this launch sets to 0 (DgCorrArgs.wctr), or null
    DgGroupArgs gr;                    // gr.nkeys > 0: blockIdx.z == 2 writes the consumer lists of k_corr2's grouped ragged blocks
    DgDenseCodeArgs dc;                // dc.B > 0: blockIdx.z == 3 builds the dense code operands (and blockIdx.z == 1 is the k_rowmean launch's)
    int32_t zsel;                      // 0: every role; 1: only the dense code operands (blockIdx.z == 3); 2: every role but them - the launch
                                       //    split in two that run on two streams (dg_api.hip, exact clamp masks on the dense grid)
};

struct DgRowmeanJob {
    const char* A;            // operand-1 blobs
    const float* bbar;        // [B][KF] mean normalised feats of operand 2
    const __bf16* bsplit;     // [B][2][KF] the same as bf16 hi / lo halves (B fragments of the row-mean MFMAs)
    const int64_t* aidx;      // batch maps (null = identity)
    const int64_t* bidx;
    float* rvec;              // [B][Ppad]
    float* rimg;              // [B] per-image sums of rvec
};
struct DgRowmeanArgs {
    DgRowmeanJob jobs[DG_MAX_NEG + 2];
    int32_t njobs, B, P, Ppad, KF, KD;
    const float* abar;           // [B][KF] mean normalised feats of operand 1
    char* stash;                 // non-null: k_corr2 FOLD - job 0's row means r also go, as -r / 2 in an fp16 pair (hi, 2048 lo), into k = 0, 1
    int32_t stash_off;           //   of position p's granule of the operand-1 blobs' C part, k-step stash_off / 1024 (all channel padding)
    const float* cs_part[2];     // ncs > 0: block x == Ppad/32 + 1 of an image also reduces the code column sums of the dense
    float* cs_out[2];            //          operands (csum[o][n][d] = sum_tiles ccolpart[o][n][tile][d]), see DgDenseArgs.code_split
    int32_t ncs;
};

// General coordinates, first launch of the forward: the jobs that depend on nothing but the call's inputs, side by side -
// blocks [0, count) draw the negatives' batch maps, the next B (depth != null) resize the depth indicators, the last 2 B
// (taps != null: gradient passes) build the inverse tap records the adjoint of sample() gathers through.

struct DgScatterSrc {
    const float* buf;      // gradient tiles [B][Ppad/32][DP/32][4][64][4] (dg_gtile_off).  raw == 1: the fused kernel's
                           // d/d(normalised operand-1 code), normalisation backward pending; raw == 0: k_gs output (final)
    const int64_t* route;  // null: image n scatters to destination n; else destination = route[n]
    int32_t gidx;          // upstream scalar index (0 intra, 1 inter, 2 neg, 3 depth)
    int32_t coords_sel;    // 0: coords1, 1: coords2
    float factor;          // constant factor (1/numel etc.)
    int32_t dest;          // 0: grad_code, 1: grad_code_pos
    int32_t raw;           // see buf
    const float* dfac;     // null, or a device scalar multiplied into the factor (the fused small-grid path: old_mean of the pair-set,
                           // which only the forward's last block knows - dg_small.hip)
    int32_t half;          // 1 (identity grid, round 6): the tiles are fp16, [B][Ppad/32][DP/32][2][64][8] - accumulator elements 8s .. 8s+7 of
                           // a lane in one 16-byte piece, the layout of the G tiles.  raw == 1: as above.  raw == 0: k_gs's output PROJECTED
                           // (dx - x <x, dx>) but not yet divided by ||c||: the consumer multiplies by xinv_dest of the destination position
                           // (bounded like the raw tiles whatever the norm of a code vector is: nothing can leave the fp16 range)
};
#define DG_MAX_SCATTER 48
struct DgScatterArgs {
    DgScatterSrc src[DG_MAX_SCATTER];
    int32_t nsrc;
    const float* coords1;
    const float* coords2;
    const float* gscal;    // [DG_OUT_COUNT] upstream gradient of the output vector (device); see dg_gscal; or null:
    const float* gtot;     // [1] upstream gradient of out[DG_OUT_TOTAL] alone (dg_corr_backward_total)
    float wtot[4];         // weights of the four loss means in the total
    float* comb[2];        // gradient tiles: combined direct sources per destination (scratch)
    char* taps;            // [2 coords sets][B] inverse tap records (dg_taps_record_bytes each)
    const char* xop;       // operand-1 blobs (C part = normalised code rows the raw sources refer to)
    const float* xinv;     // [B][Ppad] 1 / max(||code||, eps) of operand 1
    const float* xinv_dest[2];   // ... of the operand whose code map destination 0 / 1 is (half, final sources; identity grid)
    int32_t blob_bytes, blob_off_c;
    float* out[2];         // grad_code, grad_code_pos  (B,D,h,w)
    int32_t B, D, DP, h, w, S, Sh, P, Ppad, DC;   // DC = channels per block (power of two <= 32)
    int32_t debug;         // developer ablation bits (0 in production)
    int32_t dense;         // 1: identity grid (DG_IDENTITY_GRID): the adjoint of sample() is a transposed copy
    // (set by the launcher) the direct sources of k_grad_combine per destination, in source order: raw ones, then final ones
    int8_t craw[2][DG_MAX_SCATTER / 2], cfin[2][DG_MAX_SCATTER / 2];
    int8_t ncraw[2], ncfin[2];
    // ... and those with fp16 tiles (DgScatterSrc.half), which the lists above then leave out (k_combine_out)
    int8_t crawh[2][DG_MAX_SCATTER / 2], cfinh[2][DG_MAX_SCATTER / 2];
    int8_t ncrawh[2], ncfinh[2];
    int32_t routed_half;   // 1: every routed source has fp16 tiles (all or none: the launcher checks)
    int32_t taps_ready;    // 1: the forward built the tap records (dg_launch_pre_general)
    // extra z slices of the k_grad_combine launch (general coordinates): axo[j] = axd[j] + axd2[j] + axf[j][0] * (axs[j] + axs2[j]), tile
    // by tile (null terms are left out) - the fused small-grid path merges the two halves of every ROUTED streamed-side source and
    // their old_mean terms into ONE buffer here, so that the adjoint launch behind it walks one routed source per negative whatever
    // the grid (dg_small.hip).  The inputs stay as they are: a second backward sees the same.
    int32_t naxpy;
    float* axo[DG_MAX_NEG + 2];
    const float* axd[DG_MAX_NEG + 2];
    const float* axd2[DG_MAX_NEG + 2];
    const float* axs[DG_MAX_NEG + 2];
    const float* axs2[DG_MAX_NEG + 2];
    const float* axf[DG_MAX_NEG + 2];
};

#ifdef __HIPCC__
__device__ __forceinline__ float dg_src_factor(const DgScatterSrc& q) { return q.dfac ? q.factor * q.dfac[0] : q.factor; }
// effective upstream gradient of loss mean i: direct + through the weighted total
__device__ __forceinline__ float dg_gscal(const DgScatterArgs& a, int i) {
    return a.gscal ? a.gscal[i] + a.gscal[DG_OUT_TOTAL] * a.wtot[i] : a.gtot[0] * a.wtot[i];
}
#endif

// k_gs: gradient w.r.t. the STREAMED operand's code from the G tiles the fused kernel stored:
//   dS[q][:] = sum_p G[q][p] * x_R[p][:], then normalisation backward with the S code.
struct DgGsJob {
    const uint16_t* G;     // fp16 tiles [B][nt(R tile)][nt(S tile)][64][16]
    const char* Rop;       // operand blobs of the stationary operand of the producing job (P part is read)
    const char* Sop;       // operand blobs of the streamed operand (C part: x_S for the normalisation backward)
    const float* ScInv;    // [B][Ppad] 1/max(||c||,eps) of the S code operand
    const int64_t* ridx;   // batch maps of the producing job (null = identity)
    const int64_t* sidx;
    float* dS;             // gradient tiles (dg_gtile_off) out
};
struct DgGsArgs {
    DgGsJob jobs[DG_MAX_NEG + 2];
    int32_t njobs, B, P, Ppad, KF, KD;
    int32_t D;             // real code channels (<= KD)
    int32_t debug;         // developer ablation bits (0 in production)
    DgFinishArgs fin;      // the first block also reduces k_corr_main's partial sums (fin.out == null: nothing to do)
    // The depth term (depth_feature_correlation) as extra blocks of this launch (dep_blocks > 0): the G-stream blocks are
    // HBM-bound and their second round leaves block slots empty, so the depth blocks' latency chain costs next to nothing here
    // (in the fused kernel's launch they were its tail).  Their partial sums arrive inside this launch: the depth block that
    // finishes last (ticket) reduces everything to the output scalars instead of the first block.
    const char* dep_op;    // operand-1 blobs: R and S of the depth term (C and P parts)
    const float* dep_nz;   // [B][Ppad] depth indicators
    float* dep_dR;         // raw gradient tiles out (dg_gtile_off)
    float* dep_part;       // [B * dep_nrb][2] loss partial sums out
    unsigned int* dep_ticket;   // zero before the launch (k_colmean)
    float dep_shift, dep_lo, dep_hi;    // shift; clamp bounds of cd (zero_clamp / stabalize)
    int32_t dep_blocks, dep_nrb;    // B * dep_nrb blocks of 8 row tiles
};

// bytes of one inverse-tap record: off[HW+1] ints, 4P weights, 4P positions (ushort), padded to 16
__host__ __device__ inline size_t dg_taps_record_bytes(int HW, int P) {
    return (((size_t)(HW + 1) * 4 + (size_t)4 * P * 6) + 15) / 16 * 16;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a host round trip of a few microseconds: do it once per kernel (and
// again only if a larger size is ever needed).  Host threads of different devices may race benignly (same value).
#include <map>
#include <mutex>
#include <utility>
inline hipError_t dg_set_max_smem(const void* kern, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;      // (kernel, device) -> bytes granted
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(kern, dev);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[key] = bytes;
    return e;
}

// launchers (defined next to their kernels)
hipError_t dg_launch_corr(const DgCorrArgs& args, int KF, int KD, int nwaves, int mode, hipStream_t stream);
hipError_t dg_launch_corr2(const DgCorrArgs& args, int KF, int KD, hipStream_t stream);   // hipErrorNotSupported: use dg_launch_corr
bool dg_corr2_supported(const DgCorrArgs& args, int KF, int KD);
bool dg_corr2_shape_supported(int KF, int KD, int D, float lo, float hi, int Ppad, int B);
hipError_t dg_launch_gs(const struct DgGsArgs& a, const uint32_t* dep_maskbits, hipStream_t s, bool depth_only = false, bool half_out = false);   // half_out: DgScatterSrc.half tiles (KF = 384, KD = 96)   // dep_maskbits: exact clamp masks of the intra pair-set (DgJob.maskbits) or null
hipError_t dg_launch_finish(const DgFinishArgs& a, hipStream_t stream);
hipError_t dg_launch_transpose(const DgTransposeArgs& a, int B, hipStream_t s);
hipError_t dg_launch_gather(const DgGatherArgs& a, int maxK, hipStream_t s);
// Exact clamp masks of the pair-sets on small sample grids: 1[<c1_p, c2_q> >= 0] from the fp32 sampled code rows (the sign of cd
// does not depend on the normalisation), packed as one word per (S tile, R position).
hipError_t dg_launch_cd_mask(const DgCdMaskArgs& a, hipStream_t s);
// The same mask words on the dense identity grid from SPLIT fp16 operands: cd = hi.hi + (hi.lo + lo.hi) / 2048 on the fp16 MFMA with
// fp32 accumulation - the operand error drops from 2^-11 to 2^-22 relative, i.e. to the rounding noise of an fp32 dot product,
// at 3 instead of 16 times the work of the fp16 chain (k_cd_mask's fp32 MFMA runs at 1/16 of the fp16 rate).
struct DgCdMask3Args {
    const char* opR;                         // operand-1 blobs (hi = their C parts)
    const char* loR;                         // ... and the dropped parts (DgDenseCodeArgs.clo)
    const char* opS[DG_MAX_NEG + 2];         // streamed operand of pair-set t
    const char* loS[DG_MAX_NEG + 2];
    const int64_t* sidx[DG_MAX_NEG + 2];     // batch map of the streamed operand (null: identity)
    uint32_t* bits[DG_MAX_NEG + 2];          // [B][Ppad/32][Ppad] out (the format of DgCdMaskArgs.bits)
    int32_t T, B, Ppad, blob_bytes, off_c, KD;
    int32_t nsplit;                          // parts the walk over the S tiles is split in (0: the launcher decides)
};
hipError_t dg_launch_cd_mask3(const DgCdMask3Args& a, hipStream_t s);
hipError_t dg_launch_plane_sample(const DgPlaneArgs& a, hipStream_t s);
hipError_t dg_launch_colmean(const DgColmeanArgs& a, hipStream_t s);
hipError_t dg_launch_prep_dense(const DgDenseArgs& a, hipStream_t s);
hipError_t dg_launch_rowmean(const DgRowmeanArgs& a, hipStream_t s);
hipError_t dg_launch_set_stash(char* blobs, int B, int ntiles, size_t blob_bytes, int off, const float* rvec, int P, int Ppad, hipStream_t s);
hipError_t dg_launch_scatter(const DgScatterArgs& a, hipStream_t s);
hipError_t dg_launch_super_perms(const float* keys, uint64_t seed, unsigned long long* state, int count, int B, int64_t* out, hipStream_t s);
hipError_t dg_launch_salience_coords(const float* sal, int B, int H, int W, int n, const float* u_sel, const float* u_fb,
                                     float* out, hipStream_t s);
hipError_t dg_launch_simple_coords(const float* depth, int B, int H, int W, int h, int w, int n, const float* u_val,
                                   const float* u_pick, float* out, hipStream_t s);
hipError_t dg_launch_confusion(const long long* preds, const long long* target, long long count, int ncls, int nrows,
                               unsigned long long* stats, hipStream_t s);
hipError_t dg_launch_sims_nt(const float* q, const float* x, long long rows_q, long long n, int F, long long q_stride, long long x_stride,
                             float* out, long long out_stride, hipStream_t s);
hipError_t dg_launch_topk_rows(const float* vals, long long rows, long long cols, long long row_stride, int k,
                               long long* out_idx, float* out_val, hipStream_t s);
hipError_t dg_launch_pre_general(const struct DgPreArgs& a, hipStream_t s);
bool dg_small_supported(int Ppad, int KD);
hipError_t dg_launch_corr_small(const struct DgSmallArgs& a, hipStream_t s);       // the fused kernel; the call's scalars need ...
hipError_t dg_launch_small_finish(const struct DgSmallArgs& a, hipStream_t s);     // ... this one-wave launch behind it
hipError_t dg_launch_gather_rows(const struct DgGatherRowsArgs& a, hipStream_t s);
hipError_t dg_launch_lhp_points(const float* depth, int B, int H, int W, int h, int w, float factor, float* points, hipStream_t s);
hipError_t dg_launch_lhp_propagate(bool backward, const float* src, const float* points, float* stats, int B, int D, int P,
                                   float* dst, hipStream_t s);
hipError_t dg_launch_lhp_map(int mode, const float* code, const float* attn, const float* points, const float* divide, int B, int D,
                             int h, int w, int heads, float* out, float* map, hipStream_t s);
hipError_t dg_launch_lhp_map_bwd(int mode, const float* g, const float* map, const float* divide, int B, int D, int h, int w,
                                 float* gcode, hipStream_t s);
hipError_t dg_launch_rand_coords_state(unsigned long long* state, float* out, int n, hipStream_t s, float keep_p = -1.f);
// (B images in all; the first Ba from `depth`, the rest from `depth_b`)
hipError_t dg_launch_fps(const float* depth, const float* depth_b, int Ba, int B, int H, int W, int h, int w, int S, float factor,
                         float* out_coords, int32_t* out_inds, float* pooled_ws, hipStream_t s);

// ---- the segmentation head (dg_head.hip; DinoFeaturizer's cluster1 / cluster2, src/modules.py:75-88, 122-137)
struct DgHeadFwdArgs {
    const float* feat;                     // (B,C,P) fp32
    const float* w1; const float* b1;      // (D,C), (D)
    const float* w2a; const float* b2a;    // (C,C), (C)   null: projection_type "linear"
    const float* w2b; const float* b2b;    // (D,C), (D)
    const __bf16* w1_bf; const __bf16* w2a_bf; const __bf16* w2b_bf;   // bf16 copies of the three weight matrices (k_head_prep)
    const float* keep1; const float* keep2; const float* keep3;   // (B,C): 1 keep / 0 drop; null: no dropout for that use
    float scale;                           // 1/(1-p) applied where a keep mask is given
    float* code;                           // (B,D,P)
    float* feats_out;                      // (B,C,P) = f * keep3 * scale, or null
    __bf16* hidden;                        // (B,C,P) bf16: ReLU output saved for the backward, or null
    int32_t B, C, D, P;
    unsigned long long* stamps;            // developer timing stamps (null in production)
    // two passes of the featurizer in one launch (dg_head_forward_pair: img, then img_pos): images Bs.. of feat / code / feats_out
    // live in a second tensor each; d_* = (its base - the first's base) in elements - Bs images, added to the offset of those images
    int32_t Bs;
    long long d_feat, d_code, d_fo;
};
// element offset of image b of a tensor that continues in a second allocation from image Bs on (see DgHeadFwdArgs)
__host__ __device__ inline long long dg_img_off(int b, long long stride, int Bs, long long delta) { return (long long)b * stride + (b >= Bs ? delta : 0); }

struct DgHeadDhArgs {
    const float* gcode;      // (B,D,P) fp32
    const __bf16* w2bT;      // (C, DP) bf16: W2b transposed, DP = D rounded up to 32, zero padded (k_head_prep)
    const __bf16* hidden;    // (B,C,P)
    __bf16* dh;              // (B,C,P) out
    float* part_bd;          // [B * tiles][D] per-block row sums of d code (bias gradients of the output convolutions)
    float* part_b2a;         // [B * tiles][C] per-block row sums of d hidden_pre
    int32_t B, C, D, P;
    int32_t Bs; long long d_gcode;   // (pair: images Bs.. of gcode in a second tensor, as DgHeadFwdArgs)
    __bf16* gcode_bf;                // (B,D,P) out or null: d code rounded to bf16, the A2h operand of k_head_wgrad3 (P a multiple of 4)
    int32_t step_major;              // 1 (k_head_dh2 in front of k_head_wgrad3): dh and gcode_bf as [image][step of 32 positions][row][32] - the rows of a step
                                     //    contiguous, what k_head_wgrad3's DMA pieces read (row-major gave them 64-byte pieces of 16 rows: 54 against 45 us)
    float* part_w2b;                 // [blocks][D][C] or null: k_head_dh2 also forms d W2b = d code x hidden^T (both tiles are in its LDS), one partial sum per block
    unsigned long long* stamps;      // developer timing stamps (null in production)
    int32_t staged;                  // (set by the launcher) 1: hidden / d hidden through an LDS image of whole rows (P a multiple of 8)
};

struct DgHeadWgradArgs {
    const void* A; const void* Bm;     // (B, M, P), (B, N, P); fp32 or bf16 (template)
    const float* keep;                 // (B, N) or null
    float* part;                       // [splits][M][N]
    int32_t B, M, N, P, splits;
    // optional second product with the same Bm in the same launch (M2 > 0): A2 (B, M2, P) fp32, its keep mask and partial sums
    const void* A2; const float* keep_2; float* part2; int32_t M2;
    const void* A2h;                   // (B, M2, P) bf16 copy of A2 in ONE tensor (k_head_dh) or null; with it the two products run as k_head_wgrad3
    int32_t a_step_major;              // k_head_wgrad3: A and A2h are [image][step of 32 positions][row][32] (k_head_dh2 wrote them so)
    int32_t Bs; long long dA, dB, dA2;   // (pair: images Bs.. of A / Bm / A2 in second tensors, offsets in their elements; 0: one tensor)
};

hipError_t dg_launch_sampled_sumsq(const float* feats, const float* coords, const int64_t* srcidx, float* out, int B, int C, int h, int w, int S, int Sh, int accumulate, hipStream_t s);
hipError_t dg_launch_normalize_split(const float* src, int B, int C, int P, int nchunks, int chunk_c, float* const* dst, hipStream_t s);
hipError_t dg_launch_head_fwd(const DgHeadFwdArgs& a, hipStream_t s);
hipError_t dg_launch_head_prep(const float* w1, const float* w2a, const float* w2b, void* scratch, int C, int D, hipStream_t s);
// The bf16 copies of the head's weights (k_head_prep -> k_head_fwd / k_head_dh*).  The three matrices the forward multiplies with are
// stored FRAGMENT-MAJOR: [16-row block][k-step of 32][lane = 16 (k / 8 % 4) + row % 16][8 bf16] - the 1 KiB a wave reads for one A
// fragment is contiguous (whole cache lines), where row-major copies gave every lane 16 bytes of 16 different rows: 64-byte pieces,
// the rate of which set the forward's k-loop (round 6).  Rows are padded to whole blocks (cluster1 / cluster2's output convolution:
// eight blocks = 128 rows, the most the forward's waves walk), channels to the forward's padded width CP; the padding is zeros.
__host__ __device__ inline int dg_head_cp(int C) { return C <= 64 ? 64 : (C <= 128 ? 128 : (C <= 192 ? 192 : (C <= 384 ? 384 : 768))); }
struct DgHeadWeightLayout {
    int CP, KS; size_t w1, w2a, w2b, w2bT, elems;          // offsets / total in bf16 elements
    __host__ __device__ DgHeadWeightLayout(int C, int D) {
        CP = dg_head_cp(C); KS = CP / 32;
        const size_t blk = (size_t)KS * 512;                // elements of one 16-row block
        w1 = 0; w2a = 8 * blk; w2b = w2a + (size_t)(CP / 16) * blk; w2bT = w2b + 8 * blk;
        elems = w2bT + (size_t)C * ((D + 31) / 32 * 32);
    }
};
hipError_t dg_launch_head_dh(const DgHeadDhArgs& a, hipStream_t s);
int dg_head_dh_fused_blocks(int B, int C, int D, int P);      // > 0: k_head_dh2 runs this shape with that many blocks and can form d W2b on the way
hipError_t dg_launch_head_wgrad(const DgHeadWgradArgs& a, bool a_bf16, bool b_bf16, hipStream_t s);
bool dg_head_wgrad_one_pass(int M, int N, int M2, int P);      // the two products over the features as k_head_wgrad3 (one block = all rows x 128 channels)
struct DgHeadReduceJob { const float* part; float* out; float* out2; int32_t n, splits; float scale; };
struct DgHeadReduceArgs { DgHeadReduceJob jobs[6]; int32_t njobs; };
hipError_t dg_launch_head_reduce(const DgHeadReduceArgs& a, hipStream_t s);
hipError_t dg_launch_head_rowsum(const void* X, bool bf16, float* out, float* out2, int B, int R, int P, hipStream_t s, int Bs = 1 << 30, long long dX = 0);

// ---- the probes (dg_probe.hip; ClusterLookup src/modules.py:647-675, linear-probe loss src/train_segmentation.py:421-434)
struct DgClusterArgs {
    const float* x;          // (B, D, P)
    const float* clusters;   // (n, D)
    float alpha;             // NaN: hard assignment (alpha is None)
    float* inner;            // (B, n, P) out
    float* probs;            // (B, n, P) out, or null
    float* logp;             // (B, n, P) out: log_softmax(alpha * inner), or null
    float* part;             // [blocks] partial sums of sum_n probs * inner
    int32_t B, D, n, P;
};

struct DgClusterBwdArgs {
    const float* x; const float* clusters; const float* inner;   // as the forward
    const float* gloss;      // [1] upstream gradient of the loss (device)
    float alpha;
    float* dinner;           // (B, n, P) scratch out: d loss / d inner
    float* grad_x;           // (B, D, P) out or null
    float* part;             // [B * ceil(P/64)][n][D] partial sums of d loss / d normalised centres
    float* grad_clusters;    // (n, D) out
    int32_t B, D, n, P;
};

struct DgProbeCeArgs {
    const float* logits;     // (B, n, h, w) the probe's output at feature resolution
    const int64_t* label;    // (B, H, W)
    float* part;             // [B * H][2]: per label row, sum of -log p[label] over the labelled pixels, their count
    const float* gloss;      // backward: [1] upstream of the loss
    const float* total;      // backward: [2] = {loss sum, count} of the forward
    float* grad_logits;      // backward: (B, n, h, w)
    int32_t B, n, h, w, H, W;
};

hipError_t dg_launch_cluster_fwd(const DgClusterArgs& a, float* loss_out, hipStream_t s);
hipError_t dg_launch_cluster_bwd(const DgClusterBwdArgs& a, hipStream_t s);
hipError_t dg_launch_probe_ce_fwd(const DgProbeCeArgs& a, float* out3, hipStream_t s);
hipError_t dg_launch_probe_ce_bwd(const DgProbeCeArgs& a, hipStream_t s);

// Shared device/host definitions for the gfx950 DepthG correlation-loss kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/depthg_corr.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

#define DG_EPS_NORM 1e-10f  // F.normalize eps, reference src/modules.py:790
#define DG_TILE 32          // positions per MFMA tile edge (v_mfma_f32_32x32x16_bf16)

// Position permutation inside each 32-position block of the P-major code operand, chosen so that
// the B fragment of the gradient product (k order = accumulator row order of a 32x32 MFMA tile,
// cdna guide section 3 "An accumulator tile as the next MFMA's operand") is one 16-byte read:
// position pl = 16*s + 8*u + 4*hh + v is stored at (2*s + hh)*8 + 4*u + v.
__host__ __device__ inline int dg_perm32(int pl) {
    int s = pl >> 4, u = (pl >> 3) & 1, hh = (pl >> 2) & 1, v = pl & 3;
    return (2 * s + hh) * 8 + 4 * u + v;
}

// job kinds of the fused correlation kernel
enum { DG_JOB_HELPER = 0, DG_JOB_DEPTH = 1 };

// One pass of the row-stationary correlation kernel over one pair-set.
// "R" = stationary operand (its positions live on MFMA lanes / output rows of the gradient),
// "S" = streamed operand (tiles of 32 positions through LDS).
struct DgJob {
    const uint16_t* Rf;   // bf16 [B][Ppad][KF]  normalised feats, K-major   (null for DG_JOB_DEPTH)
    const uint16_t* Rc;   // fp16 [B][Ppad][KD]  normalised code,  K-major (fp16: 4x finer than bf16, range is [-1,1])
    const uint16_t* Sf;   // bf16 [B][Ppad][KF]
    const uint16_t* Sc;   // fp16 [B][Ppad][KD]
    const uint16_t* ScP;  // fp16 [B][KD][Ppad]  normalised code, P-major, dg_perm32-permuted per 32-block
    const float* rvec;    // fp32 [B][Ppad] row means a_p . bbar (indexed by operand-1 position) or null
    const float* rsum;    // fp32 [B] per-image sums of rvec over valid p (for m0) or null
    const float* nzR;     // fp32 [B][Ppad] depth indicators (DG_JOB_DEPTH)
    const float* nzS;
    const float* RcInv;   // fp32 [B][Ppad] 1/max(||c||,eps) of the R code operand (normalisation backward)
    const int64_t* ridx;  // batch index map of R operands (null = identity)
    const int64_t* sidx;  // batch index map of S operands (null = identity)
    float* dR;            // fp32 [B][Ppad][DP] gradient w.r.t. the sampled (unnormalised) R code, unit upstream; or null
    float* part;          // fp32 [blocks of this job][2] partial sums (sum clamp(cd)*(fd-shift), sum cd); or null
    float* out_cd;        // fp32 [B][P][P] (op1 position major) or null    (materialise; needs center_on_lane == 0)
    float* out_loss;      // fp32 [B][P][P] or null
    float shift;
    int32_t kind;
    int32_t center_on_lane;  // 1: R is operand 1 (rvec / nzR indexed by lane); 0: R is operand 2 (rvec by tile row)
    int32_t pad_;
};

#define DG_MAX_JOBS 24

struct DgCorrArgs {
    DgJob jobs[DG_MAX_JOBS];
    int32_t njobs;
    int32_t B, P, Ppad;
    int32_t nrb;          // row blocks per image = ceil(Ppad / (NWAVES*32))
    int32_t D;            // real code channels
    float lo, hi;         // clamp bounds
    float inv_BP;         // 1 / (B*P)
};

// ---- argument blocks of the helper kernels (one definition shared by kernels and host API)
struct DgFinishArgs {
    const float* part[DG_MAX_JOBS];  // per job: [nblk][2] partial (loss, cd) sums
    int32_t nblk[DG_MAX_JOBS];
    int32_t slot_loss[DG_MAX_JOBS];  // which output scalar the loss sum of job j adds to (-1 none)
    int32_t slot_cd[DG_MAX_JOBS];
    float scale[DG_MAX_JOBS];        // 1/numel of the tensor the job contributes to
    int32_t njobs;
    const float* nz;                 // [B][Ppad] or null
    int32_t B, P, Ppad;
    float* out;                      // [DG_OUT_COUNT]
};

struct DgGatherJob {
    const float* src;        // NHWC fp32 [B][h*w][K4]
    const float* coords;     // [B][S][S][2]
    const int64_t* srcidx;   // batch map (image n is read from src[srcidx[n]]) or null
    uint16_t* outK;          // bf16 [B][Ppad][Kpad]
    uint16_t* outP;          // bf16 [B][Kpad][Ppad] permuted (code operands) or null
    float* inv_norm;         // [B][Ppad] or null
    float* colpart;          // [B][Ppad/32][Kpad] per-tile column sums of the normalised rows or null
    int32_t K, K4, Kpad;
    int32_t fp16;            // 1: write IEEE half (code operands), 0: bf16 (feats operands)
    int32_t pad_;
};
#define DG_MAX_GATHER 20
struct DgGatherArgs {
    DgGatherJob jobs[DG_MAX_GATHER];
    int32_t njobs, B, h, w, S, P, Ppad;
};

struct DgRowmeanJob {
    const uint16_t* A;        // bf16 [B][Ppad][KF] operand-1 feats
    const float* colpart;     // [B][Ppad/32][KF] column sums of the operand-2 feats
    const int64_t* aidx;      // batch maps (null = identity)
    const int64_t* bidx;
    float* rvec;              // [B][Ppad]
    float* rsum;              // [B]
};
struct DgRowmeanArgs {
    DgRowmeanJob jobs[DG_MAX_NEG + 2];
    int32_t njobs, B, P, Ppad, KF;
};

struct DgScatterSrc {
    const float* buf;      // fp32 [B][Ppad][DP]
    const int64_t* route;  // null: image n scatters to destination n; else destination = route[n]
    int32_t gidx;          // upstream scalar index (0 intra, 1 inter, 2 neg, 3 depth)
    int32_t coords_sel;    // 0: coords1, 1: coords2
    float factor;          // constant factor (1/numel etc.)
    int32_t dest;          // 0: grad_code, 1: grad_code_pos
};
#define DG_MAX_SCATTER 32
struct DgScatterArgs {
    DgScatterSrc src[DG_MAX_SCATTER];
    int32_t nsrc;
    const float* coords1;
    const float* coords2;
    const float* gscal;    // [4] upstream gradients (device)
    float* out[2];         // grad_code, grad_code_pos  (B,D,h,w)
    int32_t B, D, DP, h, w, S, P, Ppad, DC;   // DC = channels per block (power of two <= 32)
};

// launchers (defined next to their kernels)
hipError_t dg_launch_corr(const DgCorrArgs& args, int KF, int KD, int nwaves, bool grad, hipStream_t stream);
hipError_t dg_launch_finish(const DgFinishArgs& a, hipStream_t stream);
hipError_t dg_launch_transpose(const float* src, float* dst, int B, int K, int HW, int K4, hipStream_t s);
hipError_t dg_launch_gather(const DgGatherArgs& a, int maxK, hipStream_t s);
hipError_t dg_launch_depth_nz(const float* depth, float* nz, int B, int H, int W, int S, int Ppad, hipStream_t s);
hipError_t dg_launch_rowmean(const DgRowmeanArgs& a, hipStream_t s);
hipError_t dg_launch_scatter(const DgScatterArgs& a, hipStream_t s);
hipError_t dg_launch_fps(const float* depth, int B, int H, int W, int h, int w, int S, float factor,
                         float* out_coords, int32_t* out_inds, hipStream_t s);

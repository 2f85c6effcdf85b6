// Operand preparation kernels (HBM-bound byte movers) for the correlation-loss path:
//   k_nchw_to_nhwc   (B,K,h,w) fp32 -> (B,h*w,K4) fp32, K4 = round_up(K,4), zero padded
//   k_gather_norm    sample() + norm() of the reference (src/modules.py:822-825, 789-790):
//                    bilinear gather at coords (grid_sample, border, align_corners=True),
//                    L2-normalise over channels, write the tile blobs (+ 1/norm, column sums)
//   depth_nz_image   (dg_common.h; a role of k_prep_dense / k_pre_general) depth -> F.interpolate(size=(S,S), bilinear, align_corners=True) -> norm over
//                    the single channel (src/modules.py:1261-1265): d / max(|d|, 1e-10)
//   k_rowmean        r[n][p] = a[n][p] . mean_q b[n][q]   (row means of fd for `pointwise`,
//                    src/modules.py:1236-1239 restated as a rank-1 term)
#include "dg_common.h"
#include "dg_taps.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------
// all four maps of a step (feats, feats_pos, code, code_pos) in one launch: blockIdx.y walks the 32-channel groups of the
// maps one after the other
__global__ void k_nchw_to_nhwc(const DgTransposeArgs a) {
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    int m = 0, gy = blockIdx.y;
    while (m < a.nmaps - 1 && gy >= (a.K4[m] + 31) / 32) { gy -= (a.K4[m] + 31) / 32; ++m; }
    const float* __restrict__ src = a.src[m];
    float* __restrict__ dst = a.dst[m];
    const int K = a.K[m], K4 = a.K4[m], HW = a.HW[m];
    const int k0 = gy * 32, p0 = blockIdx.x * 32;
    if (p0 >= HW) return;                            // (the grid spans the largest map)
    const int tx = threadIdx.x, ty = threadIdx.y;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int k = k0 + i, p = p0 + tx;
        t[i][tx] = (k < K && p < HW) ? src[((size_t)b * K + k) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int p = p0 + i, k = k0 + tx;
        if (p < HW && k < K4) dst[((size_t)b * HW + p) * K4 + k] = t[tx][i];
    }
}

hipError_t dg_launch_transpose(const DgTransposeArgs& a, int B, hipStream_t s) {
    int gy = 0, hw = 0;
    for (int m = 0; m < a.nmaps; ++m) { gy += (a.K4[m] + 31) / 32; hw = a.HW[m] > hw ? a.HW[m] : hw; }
    dim3 grid((hw + 31) / 32, gy, B), block(32, 8);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, block, 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------

// One (S tile, R tile) pair of the exact clamp masks (k_cd_mask below; also extra blocks of the k_gather_norm launch)
template <int NC>      // 16-byte chunks of a code row per lane: D4 <= 8 NC
__device__ __forceinline__ void cd_mask_tile(const DgCdMaskArgs& a, const int w, const int n, const int t) {
    // One wave per (S tile, R tile) pair: the 32 x 32 raw dot products on the fp32 MFMA (v_mfma_f32_32x32x2_f32: fp32 products,
    // fp32 accumulation - the sign of the reference's fp32 cd, which the fp16 MFMA of k_corr_main flips for |cd| < ~1e-3).
    // Lane (j = lane & 31, h = lane >> 5) holds the 16-byte chunks 2m + h of S row j (A) and of R row j (B); k-step (m, e) of the
    // MFMA pairs element e of chunk 2m (lanes h = 0) with element e of chunk 2m + 1 (h = 1) - the MFMA does not care which two
    // elements of the rows share a step.  A chunk past the row (odd chunk counts, NC larger than the row) is read at a clamped
    // address and zeroed.  The lane ends with column j of the tile: the bits of R position j for 16 of the 32 S positions
    // (rows (reg & 3) + 8 (reg >> 2) + 4 h - the bit order k_corr_main's epilogue reads); the halves are OR-ed across lane ^ 32.
    // (Measured at C3's shape, 3584 tile pairs: an LDS-staged fp32 FMA form 50 us, LDS-bound; rows through the scalar cache as
    // SGPR operands of v_pk_fma_f32 27 us, bound by the scalar-cache misses of each row; this form with 8-byte loads 19 us.)
    const int nt = a.Ppad >> 5, D4 = a.D4, nq = D4 >> 2;
    if (w >= nt * nt) return;
    const int st = w / nt, rt = w - st * nt;
    const int nS = a.sidx[t] ? (int)a.sidx[t][n] : n;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int ps = st * 32 + j, pr = rt * 32 + j;
    const float* srow = a.rowsS[t] + ((size_t)nS * a.P + min(ps, a.P - 1)) * D4;
    const float* rrow = a.rowsR + ((size_t)n * a.P + min(pr, a.P - 1)) * D4;
    f32x4 av[NC], bv[NC];
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        const int c = min(2 * m + h, nq - 1);
        av[m] = *reinterpret_cast<const f32x4*>(srow + 4 * c);
        bv[m] = *reinterpret_cast<const f32x4*>(rrow + 4 * c);
    }
    // (hipcc's scheduler otherwise sinks every load next to the MFMA that consumes it - the accumulator chain is serial, so it
    //  sees nothing to gain - and the wave walks 2 NC memory round trips one after the other: 14 us instead of .. at C3's shape)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NC; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) av[m][e] = 2 * m + h < nq ? av[m][e] : 0.f;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int m = 0; m < NC; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][e], bv[m][e], acc, 0, 0, 0);
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) word |= (acc[i] >= 0.f ? 1u : 0u) << ((i & 3) + 8 * (i >> 2) + 4 * h);
    word |= (uint32_t)__shfl_xor((int)word, 32);
    // padding: cd = 0 there, inside the clamp (what the fp16 path sees)
    const int nvalid = a.P - st * 32;
    if (nvalid < 32) word |= ~0u << max(nvalid, 0);
    if (pr >= a.P) word = ~0u;
    if (h == 0) a.bits[t][((size_t)n * nt + st) * a.Ppad + pr] = word;
}

// block = 256 threads = 4 waves; block handles 32 consecutive positions (one operand tile),
// wave w handles positions w, w+4, ...; lane l handles channels 4l + 256 m.  Output goes straight into
// the tile blob (dg_common.h): feats -> F part (bf16, swizzled rows), code -> C part (fp16, granule-major)
// and P part (fp16, P-major, dg_perm32 order; transposed through LDS).
template <int MAXM, int NC = 0>      // NC > 0: the launch carries the exact clamp masks (cd_mask_tile<NC>) in extra z slices
__global__ __launch_bounds__(256) void k_gather_norm(const DgGatherArgs a) {
    int zjob = (int)blockIdx.z;
    if (a.pre_blocks > 0 && zjob >= a.pre_z0) {
        // the input-only jobs of the call (k_pre_general's, except the draw): depth indicators, inverse tap records
        extern __shared__ __attribute__((aligned(16))) char gn_dyn[];
        const int lid = ((zjob - a.pre_z0) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (lid < a.pre_nz) depth_nz_image(a.pre.depth, a.pre.nz, a.pre.nzsum, lid, a.pre.dH, a.pre.dW, a.pre.Sh, a.pre.S, a.pre.Ppad);
        else if (lid < a.pre_blocks) {
            const DgTapsArgs t{a.pre.coords1, a.pre.coords2, a.pre.taps, a.pre.B, a.pre.h, a.pre.w, a.pre.S, a.pre.Sh, a.pre.P};
            const int b = lid - a.pre_nz;
            build_taps_block<256>(t, b % a.pre.B, b / a.pre.B, gn_dyn);
        }
        return;
    }
    if constexpr (NC > 0) {
        // (the mask slices FIRST: fp32-MFMA work that runs beside the memory-bound gather blocks instead of behind them)
        const int ncd = a.cd.T * a.cd_xper;
        if (zjob < ncd) {
            const int t = zjob / a.cd_xper;
            cd_mask_tile<NC>(a.cd, ((zjob - t * a.cd_xper) * (int)gridDim.x + (int)blockIdx.x) * 4 + (int)(threadIdx.x >> 6), (int)blockIdx.y, t);
            return;
        }
        zjob -= ncd;
    }
    // [KD<=128][32 positions] for the P part, rows of 40 halves (80 bytes) with the four 16-byte granules of a row XOR-ed by
    // (channel >> 4) & 3: a wave writes one position column of up to 32 channels 4 apart at a time - with rows of 64 bytes all
    // of them fell into ONE bank (3.6 M conflict cycles per launch at config 3's shape, a fifth of the kernel)
    constexpr int PTS = 40;
    __shared__ __attribute__((aligned(16))) uint16_t ptile[128 * PTS];
    __shared__ float colred[4][MAXM * 256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pt = blockIdx.x, n = blockIdx.y;
    const DgGatherJob& J = a.jobs[zjob];
    const int K4 = J.K4, Kpad = J.Kpad;
    const int ns = J.srcidx ? (int)J.srcidx[n] : n;
    const int mh = J.h, mw = J.w;                  // the map this job samples (feature and code maps may differ in size)
    const float* img = a.direct ? J.src + (size_t)n * a.P * K4 : J.src + (size_t)ns * mh * mw * K4;
    const int S = a.S, Sh = a.Sh;
    const DgBlob L(a.KF, a.KD);
    char* blob = J.blob + ((size_t)n * (a.Ppad / 32) + pt) * L.bytes;

    float colacc[MAXM][4];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) colacc[m][0] = colacc[m][1] = colacc[m][2] = colacc[m][3] = 0.f;

    // sampled rows (k_plane_sample): the wave's eight rows are loaded up front from clamped addresses - loads under `if (k < K4)`
    // are issued and waited for one by one, 24 memory round trips per wave at K = 768 (38 us at C3's shape)
    float4 pre[8][MAXM];
    if (a.direct) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = pt * 32 + wid + 4 * q;
            const float* p00 = img + (size_t)(p < a.P ? p : a.P - 1) * K4;
#pragma unroll
            for (int m = 0; m < MAXM; ++m) {
                const int k = 4 * lane + 256 * m;
                pre[q][m] = *reinterpret_cast<const float4*>(p00 + (k < K4 ? k : K4 - 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int pi = wid + 4 * q;
        const int p = pt * 32 + pi;
        float4 v[MAXM];
#pragma unroll
        for (int m = 0; m < MAXM; ++m) v[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        float inv = 0.f;
        if (p < a.P && a.direct) {
            float ss = 0.f;
#pragma unroll
            for (int m = 0; m < MAXM; ++m) {
                const int k = 4 * lane + 256 * m;
                if (k < K4) {
                    const float4 acc = pre[q][m];
                    v[m] = acc;
                    ss += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            inv = 1.f / fmaxf(sqrtf(ss), DG_EPS_NORM);
        } else if (p < a.P) {
            // output position (i, j) = (p / S, p % S) reads x = coords[n][j][i][0], y = coords[n][j][i][1]
            const int i = p / S, j = p - i * S;
            const float* c = J.coords + (((size_t)n * S + j) * Sh + i) * 2;
            float x = ((c[0] + 1.f) / 2.f) * (float)(mw - 1);
            float y = ((c[1] + 1.f) / 2.f) * (float)(mh - 1);
            x = fminf(fmaxf(x, 0.f), (float)(mw - 1));
            y = fminf(fmaxf(y, 0.f), (float)(mh - 1));
            const float x0f = floorf(x), y0f = floorf(y);
            const float wx1 = x - x0f, wy1 = y - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            const int x0 = (int)x0f, y0 = (int)y0f;
            const bool inx = x0 + 1 <= mw - 1, iny = y0 + 1 <= mh - 1;
            const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
            const float* p00 = img + ((size_t)y0 * mw + x0) * K4;
            const float* p01 = p00 + K4;
            const float* p10 = p00 + (size_t)mw * K4;
            const float* p11 = p10 + K4;
            float ss = 0.f;
#pragma unroll
            for (int m = 0; m < MAXM; ++m) {
                const int k = 4 * lane + 256 * m;
                if (k < K4) {
                    float4 t = *reinterpret_cast<const float4*>(p00 + k);
                    float4 acc = make_float4(t.x * w00, t.y * w00, t.z * w00, t.w * w00);
                    if (inx && w01 != 0.f) { t = *reinterpret_cast<const float4*>(p01 + k); acc.x += t.x * w01; acc.y += t.y * w01; acc.z += t.z * w01; acc.w += t.w * w01; }
                    if (iny && w10 != 0.f) { t = *reinterpret_cast<const float4*>(p10 + k); acc.x += t.x * w10; acc.y += t.y * w10; acc.z += t.z * w10; acc.w += t.w * w10; }
                    if (inx && iny && w11 != 0.f) { t = *reinterpret_cast<const float4*>(p11 + k); acc.x += t.x * w11; acc.y += t.y * w11; acc.z += t.z * w11; acc.w += t.w * w11; }
                    v[m] = acc;
                    ss += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            inv = 1.f / fmaxf(sqrtf(ss), DG_EPS_NORM);
        }
        // (a channel chunk of a wider map: the norm of the whole sampled vector, formed over all chunks in front of this call)
        if (J.ext_inv && p < a.P) inv = J.ext_inv[(size_t)n * a.P + p];
        // normalise, convert, store (zero rows for p >= P, zero columns for k >= K4)
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            const int k = 4 * lane + 256 * m;
            if (k < Kpad) {
                float4 u = make_float4(v[m].x * inv, v[m].y * inv, v[m].z * inv, v[m].w * inv);
                uint2 o;
                // column sums of the ROUNDED values: exactly what the MFMAs see (row means, cd means stay consistent)
                if (J.is_code) {
                    f16x4 t; t[0] = (_Float16)u.x; t[1] = (_Float16)u.y; t[2] = (_Float16)u.z; t[3] = (_Float16)u.w;
                    colacc[m][0] += (float)t[0]; colacc[m][1] += (float)t[1]; colacc[m][2] += (float)t[2]; colacc[m][3] += (float)t[3];
                    o = *reinterpret_cast<uint2*>(&t);
                    *reinterpret_cast<uint2*>(blob + L.c(pi, k >> 3) + (k & 7) * 2) = o;
                    const int pp = dg_perm32(pi);
                    const uint16_t* ob = reinterpret_cast<const uint16_t*>(&o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ptile[(k + e) * PTS + ((((pp >> 3) ^ (k >> 4)) & 3) << 3) + (pp & 7)] = ob[e];
                } else {
                    bf16x4 t; t[0] = (__bf16)u.x; t[1] = (__bf16)u.y; t[2] = (__bf16)u.z; t[3] = (__bf16)u.w;
                    colacc[m][0] += (float)t[0]; colacc[m][1] += (float)t[1]; colacc[m][2] += (float)t[2]; colacc[m][3] += (float)t[3];
                    o = *reinterpret_cast<uint2*>(&t);
                    *reinterpret_cast<uint2*>(blob + L.f(pi, k >> 3) + (k & 7) * 2) = o;
                }
            }
        }
        if (J.inv_norm && lane == 0) J.inv_norm[(size_t)n * a.Ppad + p] = inv;
    }
    if (J.colpart) {
#pragma unroll
        for (int m = 0; m < MAXM; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) colred[wid][m * 256 + 4 * lane + e] = colacc[m][e];
    }
    __syncthreads();
    if (J.colpart) {
        for (int k = tid; k < Kpad; k += 256) {
            float s = colred[0][k] + colred[1][k] + colred[2][k] + colred[3][k];
            J.colpart[((size_t)n * (a.Ppad / 32) + pt) * Kpad + k] = s;
        }
    }
    if (J.is_code) {
        // P part: channel d, granule cc = 8 permuted positions = 16 bytes
        for (int id = tid; id < Kpad * 4; id += 256) {
            const int d = id >> 2, cc = id & 3;
            uint4 val = *reinterpret_cast<const uint4*>(&ptile[d * PTS + (((cc ^ (d >> 4)) & 3) << 3)]);
            *reinterpret_cast<uint4*>(blob + L.p(d, cc)) = val;
        }
    }
}

hipError_t dg_launch_gather(const DgGatherArgs& a_in, int maxK4, hipStream_t s) {
    DgGatherArgs a = a_in;
    dim3 grid(a.Ppad / 32, a.B, a.njobs), block(256);
    int nc = 0;
    if (a.cd.T > 0) {
        // the exact clamp masks as extra z slices: one wave per (S tile, R tile) pair, 4 per block, cd_xper slices of gridDim.x blocks
        if (a.cd.D4 > 128) return hipErrorInvalidValue;
        const int nt = a.Ppad / 32, need = (nt * nt + 3) / 4;
        a.cd_xper = (need + nt - 1) / nt;
        grid.z += a.cd.T * a.cd_xper;
        nc = a.cd.D4 <= 72 ? 9 : (a.cd.D4 <= 104 ? 13 : 16);
    }
    int smem = 0;
    a.pre_nz = a.pre.depth ? a.B : 0;
    a.pre_blocks = a.pre_nz + (a.pre.taps ? 2 * a.B : 0);
    a.pre_z0 = (int)grid.z;
    if (a.pre_blocks > 0) {
        const int per = (int)(grid.x * grid.y);
        grid.z += (a.pre_blocks + per - 1) / per;
        if (a.pre.taps) smem = (int)(dg_taps_record_bytes(a.pre.h * a.pre.w, a.pre.P) + (size_t)a.pre.h * a.pre.w * 4 + 16);
    }
    auto launch = [&](auto kern) -> hipError_t {
        if (smem > 0) {
            const hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, block, smem, s, a);
        return hipGetLastError();
    };
#define DG_GN4(M_) { if (nc == 0) return launch(k_gather_norm<M_, 0>); if (nc == 9) return launch(k_gather_norm<M_, 9>); \
                     if (nc == 13) return launch(k_gather_norm<M_, 13>); return launch(k_gather_norm<M_, 16>); }
    if (maxK4 <= 256) DG_GN4(1)
    if (maxK4 <= 512) DG_GN4(2)
    if (maxK4 <= 768) DG_GN4(3)
#undef DG_GN4
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// sample() for small sample grids (the S = 11 / 12 recipes), straight from the NCHW maps: one block = one SOURCE image and 32
// channels of one map.  It reads its 32 channel planes once, coalesced, into LDS and serves every operand that samples
// that image - operand 0 / 1 at the image's own coordinates, the negatives whose batch map points at it at THEIR
// coordinates - writing sampled fp32 rows [operand][image][position][K4] (128 contiguous bytes per position and block).
// k_gather_norm (direct mode) then normalises the rows and builds the operand blobs.  Against channel-last copies of the
// whole maps + a gather with four taps per position (the general path) this moves 2 x less HBM/L2 traffic at P = 121 and
// leaves no launch whose time scales with B*h*w*C twice.
// grid (sum over maps of ceil(K4 / 32), B), block 1024, dynamic LDS 32 * (h*w + 1) floats.
#define PLANE_THREADS 1024
typedef int v4i_pl __attribute__((ext_vector_type(4)));
template <int CH>         // channels per block (32 or 16: smaller planes, more blocks per CU, their load and blend phases overlap)
__global__ __launch_bounds__(PLANE_THREADS) void k_plane_sample(const DgPlaneArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pl[];          // [CH][HW + 1], then the tap table [consumers][P][8]
    __shared__ short l_o[(DG_MAX_NEG + 2) * 64 + 2], l_n[(DG_MAX_NEG + 2) * 64 + 2];
    // XCD-aware block order: the dispatcher deals consecutive blocks round-robin over the 8 XCDs, each with its own L2.  A block writes
    // 64-byte pieces (32 channels) of its consumers' rows; the other half of each 128-byte line comes from the NEXT channel group of
    // the same image - on another XCD in dispatch order, so the halves never met in an L2 and went to memory as partial lines (82 MB
    // written for 52 MB of rows at config 3).  Logical block index = XCD-major: neighbours in (image, channel group) order share an XCD.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y;
        // (measured, three alternating runs each: config 3 (B = 32) 0.1694 -> 0.1673 ms and its WRITE_SIZE 82 -> 52 MB, config 2 (B = 16)
        //  0.1579 -> 0.1574, the config-4 shard (B = 8: one image per XCD) 0.1733 -> 0.1760 - so from two images per XCD on)
        if ((total & 7) == 0 && gridDim.y >= 16) {
            const int lin = by * gridDim.x + bx, logical = (lin & 7) * (total >> 3) + (lin >> 3);
            by = logical / (int)gridDim.x; bx = logical - by * (int)gridDim.x;
        }
    }
    const int tid = threadIdx.x, HW = a.h * a.w, b = by;
    int m = 0, gy = bx;
    while (m < 3 && gy >= (a.K4[m] + CH - 1) / CH) { gy -= (a.K4[m] + CH - 1) / CH; ++m; }
    const int K = a.K[m], K4 = a.K4[m], k0 = gy * CH, kind = m >> 1, pos_map = m & 1;
    const float* __restrict__ src = a.src[m] + ((size_t)b * K + k0) * HW;
    // planes -> LDS (rows of h*w contiguous floats; channels past K are zero)
    // (all loads of a thread are issued before the first LDS store: one memory latency per block, not one per element)
    if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // (piece i = channel i / hw4, pixel quad i % hw4: ONE division per thread and batch, then carried - a division by a run-time
        //  value is ~20 VALU instructions, and sixteen of them per thread were the largest part of this launch's instruction count)
        const int n4 = CH * HW / 4, hw4 = HW / 4;
        const int dstep = PLANE_THREADS / hw4, rstep = PLANE_THREADS - dstep * hw4;
        for (int i0 = tid; i0 < n4; i0 += PLANE_THREADS * 8) {
            f32x4 t[8];
            int ccs[8], qs[8];
            int cc = i0 / hw4, q = i0 - cc * hw4;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PLANE_THREADS;
                ccs[u] = cc; qs[u] = q;
                t[u] = i < n4 && k0 + cc < K ? *(reinterpret_cast<const f32x4*>(src) + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                cc += dstep; q += rstep;
                if (q >= hw4) { q -= hw4; ++cc; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PLANE_THREADS;
                if (i < n4) {
                    float* o = pl + ccs[u] * (HW + 1) + qs[u] * 4;
                    o[0] = t[u][0]; o[1] = t[u][1]; o[2] = t[u][2]; o[3] = t[u][3];
                }
            }
        }
    } else {
        for (int i0 = tid; i0 < CH * HW; i0 += PLANE_THREADS * 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PLANE_THREADS, cc = i / HW;
                t[u] = i < CH * HW && k0 + cc < K ? src[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PLANE_THREADS;
                if (i < CH * HW) { const int cc = i / HW; pl[cc * (HW + 1) + (i - cc * HW)] = t[u]; }
            }
        }
    }
    const int c = tid % CH, ps = tid / CH;
    const bool chan = k0 + c < K4;
    const float* plane = pl + c * (HW + 1);
    float* const taps = pl + ((CH * (HW + 1) + 3) & ~3);
    // consumers of this source image, in (operand, image) order; 64 images at a time (the batch-map entries of the first chunk
    // are loaded up front, next to the plane loads: independent loads, one latency)
    int pm0[DG_MAX_NEG];
#pragma unroll
    for (int k = 0; k < DG_MAX_NEG; ++k)
        pm0[k] = (!pos_map && k + 2 < a.nops && (tid & 63) < a.B) ? (int)a.perms[(size_t)k * a.B + (tid & 63)] : -1;
    for (int n0 = 0; n0 < a.B; n0 += 64) {
        const int nn = n0 + (tid & 63);
        int cnt = 0;
        if (pos_map) {
            if (n0 == 0 && a.nops > 1) { if (tid == 0) { l_o[0] = 1; l_n[0] = (short)b; } cnt = 1; }
        } else {
            if (n0 == 0) { if (tid == 0) { l_o[0] = 0; l_n[0] = (short)b; } cnt = 1; }
#pragma unroll
            for (int k = 0; k < DG_MAX_NEG; ++k) {
                const int o = k + 2;
                if (o >= a.nops) break;
                const int src_img = n0 == 0 ? pm0[k] : (nn < a.B ? (int)a.perms[(size_t)k * a.B + nn] : -1);
                const unsigned long long hit = __ballot(src_img == b);
                if (tid == 0) {
                    int k = cnt;
                    for (unsigned long long mm = hit; mm; mm &= mm - 1) { l_o[k] = (short)o; l_n[k] = (short)(n0 + __builtin_ctzll(mm)); ++k; }
                }
                cnt += __popcll(hit);
            }
        }
        __syncthreads();                       // (also: the planes are in LDS)
        // consumers in batches of a.tap_consumers: first the tap table of every (consumer, position) of the batch - pixel offset
        // + the four bilinear weights, computed ONCE, not per channel lane - then the channel lanes blend
        for (int e0 = 0; e0 < cnt; e0 += a.tap_consumers) {
            const int ne = min(a.tap_consumers, cnt - e0);
            for (int i = tid; i < ne * a.P; i += PLANE_THREADS) {
                const int e = i / a.P, p = i - e * a.P;
                const int o = l_o[e0 + e], n = l_n[e0 + e];
                const float* coords = o == 0 ? a.coords1 : a.coords2;
                // output position (i, j) = (p / S, p % S) reads x = coords[n][j][i][0], y = coords[n][j][i][1]  (k_gather_norm)
                const int ii = p / a.S, j = p - ii * a.S;
                const float* cc = coords + (((size_t)n * a.S + j) * a.Sh + ii) * 2;
                float x = ((cc[0] + 1.f) / 2.f) * (float)(a.w - 1);
                float y = ((cc[1] + 1.f) / 2.f) * (float)(a.h - 1);
                x = fminf(fmaxf(x, 0.f), (float)(a.w - 1));
                y = fminf(fmaxf(y, 0.f), (float)(a.h - 1));
                const float x0f = floorf(x), y0f = floorf(y);
                const float wx1 = x - x0f, wy1 = y - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                const int x0 = (int)x0f, y0 = (int)y0f;
                const bool inx = x0 + 1 <= a.w - 1, iny = y0 + 1 <= a.h - 1;
                // a tap that border padding never reads gets weight 0 and the address of the first tap
                float* tt = taps + (size_t)i * 8;
                tt[0] = wy0 * wx0;
                tt[1] = inx ? wy0 * wx1 : 0.f;
                tt[2] = iny ? wy1 * wx0 : 0.f;
                tt[3] = inx && iny ? wy1 * wx1 : 0.f;
                const int pix = y0 * a.w + x0, dx = inx ? 1 : 0, dy = iny ? a.w : 0;
                reinterpret_cast<int*>(tt)[4] = pix;                 // the four taps' pixel indices (one 16-byte read in the blend)
                reinterpret_cast<int*>(tt)[5] = pix + dx;
                reinterpret_cast<int*>(tt)[6] = pix + dy;
                reinterpret_cast<int*>(tt)[7] = pix + dy + dx;
            }
            __syncthreads();
            // the blend: the launch is bound by its INSTRUCTION count (SQ counters at config 3's shape: 55 VALU + 37 SALU wave
            // instructions per 64 outputs - 44 us of VALU issue alone), so the loop carries its pointers instead of recomputing
            // them and reads all four taps unconditionally (a tap border padding never reads has weight 0 and the first tap's
            // address; three branches on the weights cost more than the LDS reads they saved)
            for (int e = 0; e < ne; ++e) {
                const int o = l_o[e0 + e], n = l_n[e0 + e];
                constexpr int PSTEP = PLANE_THREADS / CH;
                float* rows = a.rows[o][kind] + ((size_t)n * a.P + ps) * K4 + k0 + c;
                const float* tt = taps + ((size_t)e * a.P + ps) * 8;
                const int rstep = PSTEP * K4;
                if (chan && kind == 0 && a.feats_bf16) {
                    // feature rows for the fused small-grid kernel (dg_small.hip): bf16 - half the bytes written here and read there;
                    // that kernel multiplies bf16 rows anyway and takes its norms from the values it multiplies
                    __bf16* rows16 = reinterpret_cast<__bf16*>(a.rows[o][0]) + ((size_t)n * a.P + ps) * K4 + k0 + c;
#pragma unroll 2
                    for (int p = ps; p < a.P; p += PSTEP, tt += PSTEP * 8, rows16 += rstep) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(tt);
                        const v4i_pl ix = *reinterpret_cast<const v4i_pl*>(tt + 4);
                        float acc = plane[ix[0]] * w[0];
                        acc = fmaf(plane[ix[1]], w[1], acc);
                        acc = fmaf(plane[ix[2]], w[2], acc);
                        acc = fmaf(plane[ix[3]], w[3], acc);
                        *rows16 = (__bf16)acc;
                    }
                } else if (chan) {
#pragma unroll 2
                    for (int p = ps; p < a.P; p += PSTEP, tt += PSTEP * 8, rows += rstep) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(tt);
                        const v4i_pl ix = *reinterpret_cast<const v4i_pl*>(tt + 4);
                        float acc = plane[ix[0]] * w[0];
                        acc = fmaf(plane[ix[1]], w[1], acc);
                        acc = fmaf(plane[ix[2]], w[2], acc);
                        acc = fmaf(plane[ix[3]], w[3], acc);
                        *rows = acc;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Exact clamp masks on small sample grids (the S = 11 / 12 recipes).  The gradient of the loss is discontinuous in cd (the
// factor 1[cd >= 0] of zero_clamp); with fp16 code operands a fraction ~5e-4 of the elements has |cd| below the operand rounding
// and gets the other sign - an incoherent 1.2e-2 relative error of the code gradients.  The sign of cd is the sign of the RAW
// dot product of the two sampled fp32 code rows (norm() divides by positive numbers), and at P = 121 / 144 positions that is half
// a GFLOP of fp32 for the whole batch: one block per (S tile, image, pair-set) forms the 32 x P dot products in fp32 and packs
// them as one word per (S tile, R position) - bit i = position 32 tile + i - which the fused kernel reads instead of comparing.
template <int NC>
__global__ __launch_bounds__(256) void k_cd_mask(const DgCdMaskArgs a) {
    cd_mask_tile<NC>(a, blockIdx.x * 4 + (threadIdx.x >> 6), blockIdx.y, blockIdx.z);
}
// Exact clamp masks on the dense identity grid from split fp16 operands (DgCdMask3Args).  One block = 8 waves = 8 consecutive R
// tiles of one (pair-set, image); the block walks the S tiles, whose hi / lo code rows (the first 2 NKC granule rows of the C
// part and of its companion) go through two LDS buffers, register-staged two tiles ahead.  Per S tile and wave 3 NKC fp16 MFMAs
// into ONE accumulator: hi.(2048 hi) + hi.lo' + lo'.hi = 2048 cd (lo' = 2048 lo, so that no operand is a fp16 subnormal; 2048 hi is
// exact), with the rounding noise of an fp32 dot product (operands exact to 2^-22).  The 16 signs of a lane are shifted into a
// word one v_alignbit each (first version: a compare, a select and an or per element - the launch was bound by its VALU count).
// Output: the word format of k_cd_mask (bit i of word (S tile, R position) = S position 32 tile + i).
typedef int v4i_m3 __attribute__((ext_vector_type(4)));
// W = waves (R tiles) per block: a divisor of the tile count where there is one in 5..8 (25 tiles = 5 x 5, 98 = 14 x 7): with eight
// waves and 25 tiles every fourth block ran ONE wave through the whole walk (round 5: 78 us for 33 us of MFMAs at the headline).
template <int NKC, int W>
__global__ __launch_bounds__(64 * W) void k_cd_mask3(const DgCdMask3Args a) {
    constexpr int NT = 64 * W;
    using v4i = v4i_m3;
    constexpr int HB = 2 * NKC * 512;                    // bytes of the hi (or lo) rows of one tile that the chain reads
    constexpr int NPC = 2 * HB / 16;                     // 16-byte pieces of one staged tile (hi then lo)
    __shared__ __attribute__((aligned(16))) char buf[2][2 * HB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    const int nt = a.Ppad >> 5, n = blockIdx.y, t = blockIdx.z;
    // blockIdx.x = (group of W R tiles) * nsplit + part: a block walks the S tiles [s0, s1) of its part (nsplit > 1: shorter blocks, so
    // that the launch's last round of blocks is not half empty - 1120 blocks of 25 tiles on 768 block slots at the headline)
    const int nsplit = a.nsplit > 0 ? a.nsplit : 1;
    const int rgrp = (int)blockIdx.x / nsplit, part = (int)blockIdx.x - rgrp * nsplit;
    const int per = (nt + nsplit - 1) / nsplit, s0 = part * per, s1 = min(nt, s0 + per);
    if (s0 >= s1) return;
    const int rt = rgrp * W + wid;
    const bool act = rt < nt;
    const int nS = a.sidx[t] ? (int)a.sidx[t][n] : n;
    const size_t lo_tile = (size_t)a.KD * 64;            // bytes of one tile's lo part
    // stationary fragments: granule 2k + h of R position r, hi and lo
    v4i Rh[NKC], Rl[NKC], Rs[NKC];                     // Rs = 2048 Rh
    {
        const char* rb = a.opR + ((size_t)n * nt + (act ? rt : 0)) * a.blob_bytes + a.off_c;
        const char* rl = a.loR + ((size_t)n * nt + (act ? rt : 0)) * lo_tile;
#pragma unroll
        for (int k = 0; k < NKC; ++k) {
            Rh[k] = *reinterpret_cast<const v4i*>(rb + ((2 * k + h) * 32 + r) * 16);
            Rl[k] = *reinterpret_cast<const v4i*>(rl + ((2 * k + h) * 32 + r) * 16);
            f16x8 hs = __builtin_bit_cast(f16x8, Rh[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) hs[e] = hs[e] * (_Float16)2048.f;
            Rs[k] = __builtin_bit_cast(v4i, hs);
        }
    }
    const char* const sh = a.opS[t] + (size_t)nS * nt * a.blob_bytes + a.off_c;
    const char* const sl = a.loS[t] + (size_t)nS * nt * lo_tile;
    // staging registers of two tiles: the loads of tile st + 2 are requested while tile st is worked on and tile st + 1 waits in
    // the other pair - one tile of look-ahead left the block waiting for memory (83 us for 33 us of MFMAs at the headline)
    v4i sa0, sa1 = v4i{0, 0, 0, 0}, sb0, sb1 = v4i{0, 0, 0, 0};
    auto piece_src = [&](int st, int pc) -> const char* {          // piece pc of tile st: hi rows first, then lo rows
        return pc < HB / 16 ? sh + (size_t)st * a.blob_bytes + pc * 16 : sl + (size_t)st * lo_tile + (pc - HB / 16) * 16;
    };
    auto fetch = [&](int st, v4i& p0, v4i& p1) {
        const int sc = st < s1 ? st : s1 - 1;                      // (past the end: a harmless re-load)
        p0 = *reinterpret_cast<const v4i*>(piece_src(sc, tid));
        if (tid + NT < NPC) p1 = *reinterpret_cast<const v4i*>(piece_src(sc, tid + NT));
    };
    auto stash = [&](int b, const v4i& p0, const v4i& p1) {
        *reinterpret_cast<v4i*>(buf[b] + tid * 16) = p0;
        if (tid + NT < NPC) *reinterpret_cast<v4i*>(buf[b] + (tid + NT) * 16) = p1;
    };
    static_assert(NPC <= 2 * NT, "two pieces per thread");
    uint32_t* const out = a.bits[t] + (size_t)n * nt * a.Ppad + (act ? rt : 0) * 32 + r;
    auto compute = [&](int st) {
        if (!act) return;
        const char* tile = buf[(st - s0) & 1];
        f32x16 a0 = f32x16{};
#pragma unroll
        for (int k = 0; k < NKC; ++k) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(tile + ((2 * k + h) * 32 + r) * 16);
            const f16x8 al = *reinterpret_cast<const f16x8*>(tile + HB + ((2 * k + h) * 32 + r) * 16);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, Rs[k]), a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, Rl[k]), a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, Rh[k]), a0, 0, 0, 0);
        }
        // sign bits, element 15 first: neg bit i = sign of element i (the sum starts at +0 and +0 + -0 = +0: never -0)
        uint32_t neg = 0;
#pragma unroll
        for (int i = 15; i >= 0; --i) neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(a0[i]), 31);
        // element i -> bit (i & 3) + 8 (i >> 2) + 4 h: the four nibbles move to the low halves of the four bytes
        uint32_t word = (neg & 0xFu) | ((neg & 0xF0u) << 4) | ((neg & 0xF00u) << 8) | ((neg & 0xF000u) << 12);
        word = (~word & 0x0F0F0F0Fu) << (4 * h);
        word |= (uint32_t)__shfl_xor((int)word, 32);
        if (h == 0) out[(size_t)st * a.Ppad] = word;
    };
    fetch(s0, sa0, sa1);
    stash(0, sa0, sa1);
    fetch(s0 + 1, sa0, sa1);
    fetch(s0 + 2, sb0, sb1);
    // (LDS-only barriers: __syncthreads() is also a fence of global memory - s_waitcnt vmcnt(0) in front of it - and so waited for
    //  the two tiles of look-ahead at every tile: 80 us for 33 us of MFMAs at the headline)
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    for (int st = s0; st < s1; st += 2) {
        // buf[0]: tile st; (sa): tile st + 1; (sb): tile st + 2
        lds_barrier();
        compute(st);
        stash(1, sa0, sa1);
        fetch(st + 3, sa0, sa1);
        if (st + 1 >= s1) break;
        lds_barrier();
        compute(st + 1);
        stash(0, sb0, sb1);
        fetch(st + 4, sb0, sb1);
    }
}
hipError_t dg_launch_cd_mask3(const DgCdMask3Args& a_in, hipStream_t s) {
    if (a_in.KD != 96) return hipErrorInvalidValue;
    const int nt = a_in.Ppad / 32;
    // (W = 8 and one part measured fastest at the headline - 80.5 us against 90 with W = 5 and two parts, round 6: the staging of the
    //  S tiles per R tile, not the idle waves of the last block, sets the pace; the other forms stay selectable in developer builds)
    int w = 8;
    DgCdMask3Args a = a_in;
    if (a.nsplit <= 0) a.nsplit = 1;
#ifdef DG_DEVTOOLS
    if (const char* e = getenv("DG_MASK3_SPLIT")) a.nsplit = atoi(e);
    if (const char* e = getenv("DG_MASK3_W")) w = atoi(e);
    if (w < 5 || w > 8) w = 8;
#endif
    const dim3 grid(((nt + w - 1) / w) * a.nsplit, a.B, a.T);
    switch (w) {
        case 5: hipLaunchKernelGGL((k_cd_mask3<5, 5>), grid, dim3(320), 0, s, a); break;
        case 6: hipLaunchKernelGGL((k_cd_mask3<5, 6>), grid, dim3(384), 0, s, a); break;
        case 7: hipLaunchKernelGGL((k_cd_mask3<5, 7>), grid, dim3(448), 0, s, a); break;
        default: hipLaunchKernelGGL((k_cd_mask3<5, 8>), grid, dim3(512), 0, s, a); break;
    }
    return hipGetLastError();
}

hipError_t dg_launch_cd_mask(const DgCdMaskArgs& a, hipStream_t s) {
    if (a.D4 > 128) return hipErrorInvalidValue;
    const int nt = a.Ppad / 32;
    const dim3 grid((nt * nt + 3) / 4, a.B, a.T);     // one wave per (S tile, R tile)
    if (a.D4 <= 72)       hipLaunchKernelGGL(k_cd_mask<9>, grid, dim3(256), 0, s, a);       // D = 70 (the headline width)
    else if (a.D4 <= 104) hipLaunchKernelGGL(k_cd_mask<13>, grid, dim3(256), 0, s, a);      // D = 90, 100
    else                  hipLaunchKernelGGL(k_cd_mask<16>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t dg_launch_plane_sample(const DgPlaneArgs& a, hipStream_t s) {
    constexpr int CH = 32;
    int gx = 0;
    for (int m = 0; m < 4; ++m) gx += (a.K4[m] + CH - 1) / CH;
    // planes + the tap table of a batch of consumers (32 bytes per position), as many consumers as the LDS takes (<= 8)
    const int plane_bytes = ((CH * (a.h * a.w + 1) + 3) & ~3) * 4;
    DgPlaneArgs a2 = a;
    a2.tap_consumers = (150 * 1024 - plane_bytes) / (a.P * 32);
    if (a2.tap_consumers < 1) return hipErrorInvalidValue;
    if (a2.tap_consumers > 8) a2.tap_consumers = 8;
    const int smem = plane_bytes + a2.tap_consumers * a.P * 32;
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_plane_sample<CH>), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_plane_sample<CH>, dim3(gx, a.B), dim3(PLANE_THREADS), smem, s, a2);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Dense identity grid (S == h == w, coords = the pixel centres): sample() is an exact spatial transpose
// (out[b,:,i,j] = t[b,:,j,i], reference quirk Q3), so the feats operand is built straight from the NCHW map: no
// channel-last copy, no bilinear taps.  POSITION ORDER on this grid: position p = pixel index y*w + x (the reference numbers the
// same sample x*S + y; the loss only ever sums over positions, and the order is the same for every operand, the depth indicators
// and the adjoint - the un-reduced outputs of dg_corr_materialize are written at the reference's index).  One block per (tile of
// 32 positions, image): per channel the tile is 128 contiguous, 128-byte aligned bytes of the plane; the block normalises its 32
// positions and writes the tile's whole F part (24 KB at C = 384, contiguous).
// (Round 2 kept the reference's order: one block per source row, 112-byte segments in, rows of 28 different tiles out.)
template <int MAXU, bool FK>      // FK: the deferred Dropout2d of the source's channels (its own instantiation: 28 registers)
     // channels per thread = KF / 8 <= MAXU
__device__ __forceinline__ void prep_dense_feats(const DgDenseArgs& a, float* sl, int tile, int n, int o) {
    // thread (xl = tid & 31, k0 = tid >> 5) owns position xl of the tile and the channels k0 + 8u: all of them are loaded in one
    // batch (KF/8 <= 96 loads in flight per thread), the squared norm is reduced over the 8 threads of a position through LDS,
    // and only the NORMALISED bf16 row tile goes to LDS (half the bytes of an fp32 stage -> twice the blocks per CU).
    const int tid = threadIdx.x, xl = tid & 31, x = tile * 32 + xl, k0 = tid >> 5;
    const int K = a.K, KF = a.KF, w = a.P;           // (the tile walks the flattened plane: a "row" of P pixels)
    const int xb = tile;
    const int wseg = min(32, a.P - tile * 32);      // positions of this tile
    const int RS = KF * 2 + 16;                     // LDS row stride in bytes: 16-byte aligned rows (granule reads), 2-way writes at worst
    const DgBlob L(a.KF, a.KD);
    const float* src = a.src[o] + (size_t)n * K * a.h * a.w + x;
    char* tb = reinterpret_cast<char*>(sl);         // [32][RS] bf16 rows
    float* red = reinterpret_cast<float*>(tb + 32 * RS);   // [8][32] partial squared norms
    const int nu = KF >> 3;
    float t[MAXU];
    float ss = 0.f;
    // Dropout2d of this source's channels (nn.Dropout2d in DinoFeaturizer.forward, src/modules.py:122-137, deferred to here by
    // dg_corr_forward_masked): x * (keep * scale), the fp32 product its producer would have stored - same bits downstream.  The
    // flags are requested in front of the map's values so that both batches of loads are in flight together.
    float f[FK ? MAXU : 1];
    const float* const kp = (FK && a.fkeep[o]) ? a.fkeep[o] + (size_t)n * K : nullptr;
    if constexpr (FK) {
        if (kp) {
#pragma unroll
            for (int u = 0; u < MAXU; ++u) { const int k = k0 + 8 * u; f[u] = (u < nu && k < K) ? kp[k] : 0.f; }
        }
    }
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        const int k = k0 + 8 * u;
        t[u] = (u < nu && x < w && k < K) ? src[(size_t)k * a.h * a.w] : 0.f;
    }
    if constexpr (FK) {
        if (kp) {
#pragma unroll
            for (int u = 0; u < MAXU; ++u) t[u] = t[u] * (f[u] * a.fscale);
        }
    }
#pragma unroll
    for (int u = 0; u < MAXU; ++u) ss = fmaf(t[u], t[u], ss);
    red[k0 * 32 + xl] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += red[j * 32 + xl];
    const float inv = a.unit ? 1.f : 1.f / fmaxf(sqrtf(ss), DG_EPS_NORM);     // (DG_FEATS_UNIT: a chunk of a vector normalised over all of its channels)
    if (x < w) {
#pragma unroll
        for (int u = 0; u < MAXU; ++u)
            if (u < nu) *reinterpret_cast<__bf16*>(tb + xl * RS + (k0 + 8 * u) * 2) = (__bf16)(t[u] * inv);
    }
    __syncthreads();
    if (DG_DBG(a.debug) & 32) return;                       // (ablation: loads + normalisation only)
    // blob rows: position p = y*w + x; the lanes of a row run along K
    const int pieces = KF / 8;                      // granules (16 bytes) per row
    for (int id = tid; id < wseg * pieces; id += 256) {
        const int xx = id / pieces, g = id - xx * pieces;
        const uint4 v = *reinterpret_cast<const uint4*>(tb + xx * RS + g * 16);
        const int p = xb * 32 + xx;                         // position = pixel index
        char* blob = a.blob[o] + ((size_t)n * (a.Ppad / 32) + (p >> 5)) * L.bytes;
        *reinterpret_cast<uint4*>(blob + L.f(p & 31, g)) = v;
    }
    if (DG_DBG(a.debug) & 64) return;                       // (ablation: no column sums)
    // per-source-row column sums of the normalised (bf16-rounded, i.e. exactly what the MFMA sees) rows
    for (int k = tid; k < KF; k += 256) {
        float cs = 0.f;
        for (int xx = 0; xx < wseg; ++xx) cs += (float)*reinterpret_cast<const __bf16*>(tb + xx * RS + k * 2);
        a.colpart[o][((size_t)n * (a.Ppad / 32) + tile) * KF + k] = cs;      // one partial per tile
    }
    // zero rows of the ragged last tile (positions P .. Ppad-1), once per image
    if (tile == a.Ppad / 32 - 1) {
        for (int idx = tid; idx < (a.Ppad - a.P) * (KF / 4); idx += 256) {
            const int p = a.P + idx / (KF / 4), k = (idx % (KF / 4)) * 4;
            char* blob = a.blob[o] + ((size_t)n * (a.Ppad / 32) + (p >> 5)) * L.bytes;
            *reinterpret_cast<uint2*>(blob + L.f(p & 31, k >> 3) + (k & 7) * 2) = make_uint2(0u, 0u);
        }
    }
}

// Code operand on the identity grid: one block per DENSE_TPB consecutive tiles (position = pixel index: a run of positions is a
// run of pixels of every channel plane).
// L2-normalise over the D channels (norm(), src/modules.py:789-790; squared norms reduced from registers), keep the
// NORMALISED fp16 tile in LDS and write from it the C part (K-major granules), the P part (position-major granules in
// dg_perm32 order), 1/max(||c||, eps) and the per-tile column sums.  Same roundings as k_gather_norm.
#ifndef DENSE_TPB
#define DENSE_TPB 2
#define DENSE_CODE_PAIRS 192      // (source row, column) pairs of one block: S * (columns touched) <= 192
#endif
template <int UN>        // channels per thread: D <= 4 * UN
__device__ __forceinline__ void prep_dense_code(const DgDenseArgs& a, float* sl, int tb, int n, int o) {
    const int tid = threadIdx.x, ps = tid & 63, kg = tid >> 6;
    const int KD = a.KD, D = a.D, HW = a.h * a.w, nt = a.Ppad / 32;
    const int RS = (KD + 4) * 2;                       // LDS row stride (bytes): 8-byte aligned rows, odd multiple of 8
    const int t0 = tb * DENSE_TPB, ntile = min(DENSE_TPB, nt - t0);
    const int p0 = t0 * 32, np = ntile * 32, pend = min(p0 + np, a.P);

    char* xt = reinterpret_cast<char*>(sl);            // [np][RS] normalised fp16 rows
    float* red = reinterpret_cast<float*>(xt + DENSE_TPB * 32 * RS);      // [4][DENSE_CODE_PAIRS] partial squared norms
    float* inv = red + 4 * DENSE_CODE_PAIRS;                               // [np]
    const DgBlob L(a.KF, a.KD);
    // zero tile (padding channels D..KD-1, positions P..Ppad-1 of the ragged last tile), 8 bytes per store
    for (int id = tid; id < np * (RS / 8); id += 256) reinterpret_cast<uint2*>(xt)[id] = make_uint2(0u, 0u);
    for (int pos = tid; pos < np; pos += 256) inv[pos] = 0.f;
    const float* src = a.code[o] + (size_t)n * D * HW;
    // lanes walk the positions, the four waves split the channels (k = wave + 4u); all loads of a thread in flight at once
    constexpr int NJ = DENSE_CODE_PAIRS / 64;            // <= 3 pairs per lane, all their channels in one batch
    float t[NJ][UN];
    int pos[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int pl = ps + 64 * j;                        // position inside the block's tiles = pixel p0 + pl (position = pixel index)
        const bool ok = pl < np && p0 + pl < pend;
        pos[j] = ok ? pl : -1;
        const float* sp = src + (ok ? p0 + pl : 0);
#pragma unroll
        for (int u = 0; u < UN; ++u) t[j][u] = (ok && kg + 4 * u < D) ? sp[(size_t)(kg + 4 * u) * HW] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < UN; ++u) ss = fmaf(t[j][u], t[j][u], ss);
        red[kg * DENSE_CODE_PAIRS + ps + 64 * j] = ss;
    }
    __syncthreads();
    if (DG_DBG(a.debug) & 8) return;                                  // (ablation: loads only)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (pos[j] < 0) continue;
        const int pair = ps + 64 * j;
        const float ss = red[pair] + red[DENSE_CODE_PAIRS + pair] + red[2 * DENSE_CODE_PAIRS + pair] + red[3 * DENSE_CODE_PAIRS + pair];
        const float iv = 1.f / fmaxf(sqrtf(ss), DG_EPS_NORM);
        if (kg == 0) { inv[pos[j]] = iv; }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (kg + 4 * u < D) *reinterpret_cast<_Float16*>(xt + pos[j] * RS + (kg + 4 * u) * 2) = (_Float16)(t[j][u] * iv);
    }
    __syncthreads();
    for (int pos2 = tid; pos2 < np; pos2 += 256) a.inv_norm[o][(size_t)n * a.Ppad + p0 + pos2] = inv[pos2];
    if (DG_DBG(a.debug) & 16) return;                                 // (ablation: no output phases)
    char* blob0 = a.blob[o] + ((size_t)n * nt + t0) * L.bytes;
    const int GD = KD / 8;
    for (int id = tid; id < ntile * GD * 32; id += 256) {     // C part: granule g of position qq of tile tl
        const int tl = id / (GD * 32), rem = id - tl * (GD * 32), g = rem >> 5, qq = rem & 31;
        const char* row = xt + (tl * 32 + qq) * RS + 16 * g;
        uint4 v;
        const uint2 lo = *reinterpret_cast<const uint2*>(row), hi = *reinterpret_cast<const uint2*>(row + 8);
        v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
        *reinterpret_cast<uint4*>(blob0 + (size_t)tl * L.bytes + L.c(qq, g)) = v;
    }
    for (int id = tid; id < ntile * 4 * KD; id += 256) {      // P part: granule cc of channel d = slots 8cc .. 8cc+7
        const int tl = id / (4 * KD), rem = id - tl * (4 * KD), cc = rem / KD, d = rem - cc * KD;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pl = 16 * (cc >> 1) + 8 * ((e >> 2) & 1) + 4 * (cc & 1) + (e & 3);   // dg_perm32(pl) == 8 cc + e
            v[e] = *reinterpret_cast<const _Float16*>(xt + (tl * 32 + pl) * RS + d * 2);
        }
        *reinterpret_cast<f16x8*>(blob0 + (size_t)tl * L.bytes + L.p(d, cc)) = v;
    }
    for (int id = tid; id < ntile * KD; id += 256) {          // per-tile column sums (for the cd means)
        const int tl = id / KD, d = id - tl * KD;
        float cs = 0.f;
#pragma unroll 8
        for (int qq = 0; qq < 32; ++qq) cs += (float)*reinterpret_cast<const _Float16*>(xt + (tl * 32 + qq) * RS + d * 2);
        a.ccolpart[o][((size_t)n * nt + t0 + tl) * KD + d] = cs;
    }
}

// ---- the code operands from whole channel planes (DgDenseArgs.code_split) ---------------------------------------------------
// prep_dense_code reads, per channel and source row, the 3-4 pixels of its two tiles' columns: 16-byte pieces of 128-byte
// lines, every line requested by eight blocks - the address path, not HBM, bounded it (15 of its 23 us alone were the loads
// of 14 MB).  Split in two along the only dependency there is:
//   prep_dense_code_norms   (k_prep_dense launch) one thread per pixel walks the D planes (wave = 64 consecutive pixels of a
//                           plane = two lines) and writes 1/max(||c||, eps) at the pixel's position p = x*S + y
//   dense_code_planes       (k_colmean launch)    one block per (image, operand, 8 channels): reads its 8 planes once, scales,
//                           rounds to fp16, transposes through LDS ([position][8 channels] = the C part's granule) and writes
//                           granule g of the C part (512 contiguous bytes per tile), the 8 channels' P-part granules (128-byte
//                           runs) and their per-tile column sums
// Same operations in the same order as prep_dense_code (four fma chains over the channels k = kg + 4u, summed
// ((s0 + s1) + s2) + s3; value * inv rounded to fp16; column sums over the 32 positions in order): bit-identical operands.
template <int UN>
__device__ __forceinline__ void prep_dense_code_norms(const DgDenseArgs& a, int xb, int n, int o) {
    const int HW = a.h * a.w, i = xb * 256 + threadIdx.x;
    float* inv = a.inv_norm[o] + (size_t)n * a.Ppad;
    if (xb == 0)
        for (int p = a.P + threadIdx.x; p < a.Ppad; p += 256) inv[p] = 0.f;          // positions of the ragged last tile
    if (i >= HW) return;
    const float* src = a.code[o] + (size_t)n * a.D * HW + i;
    float t[4 * UN];
#pragma unroll
    for (int k = 0; k < 4 * UN; ++k) t[k] = k < a.D ? src[(size_t)k * HW] : 0.f;
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) ss[kg] = fmaf(t[kg + 4 * u], t[kg + 4 * u], ss[kg]);
    const float tot = ss[0] + ss[1] + ss[2] + ss[3];
    inv[i] = 1.f / fmaxf(sqrtf(tot), DG_EPS_NORM);          // position = pixel index
}

__device__ __forceinline__ void dense_code_planes(const DgDenseCodeArgs& a, char* xt, int n, int o, int g) {
    const int tid = threadIdx.x, HW = a.h * a.w, nt = a.Ppad / 32, KD = a.KD;
    const DgBlob L(a.KF, a.KD);
    const int nch = min(8, a.D - 8 * g);                  // live channels of this group (<= 0: padding group, zeros)
    char* const xl = xt + (size_t)a.Ppad * 16;           // (clo wanted: the dropped parts, same [position][8 channels] image)
    const bool want_lo = a.clo[o] != nullptr;
    for (int p = a.P + tid; p < a.Ppad; p += 256) {
        reinterpret_cast<uint4*>(xt)[p] = make_uint4(0u, 0u, 0u, 0u);
        if (want_lo) reinterpret_cast<uint4*>(xl)[p] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (nch <= 0) {
        for (int p = tid; p < a.P; p += 256) {
            reinterpret_cast<uint4*>(xt)[p] = make_uint4(0u, 0u, 0u, 0u);
            if (want_lo) reinterpret_cast<uint4*>(xl)[p] = make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
        const float* src = a.code[o] + ((size_t)n * a.D + 8 * g) * HW;
        const float* inv = a.inv_norm[o] + (size_t)n * a.Ppad;
        for (int i0 = 0; i0 < HW; i0 += 1024) {           // four pixels per thread and round: 32 loads in flight
            float t[4][8], iv[4];
            int pp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + tid + 256 * j;
                const bool ok = i < HW;
                pp[j] = ok ? i : -1;                          // position = pixel index
                iv[j] = ok ? inv[pp[j]] : 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) t[j][c] = (ok && c < nch) ? src[(size_t)c * HW + i] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (pp[j] < 0) continue;
                f16x8 v;
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = c < nch ? (_Float16)(t[j][c] * iv[j]) : (_Float16)0.f;
                *reinterpret_cast<f16x8*>(xt + pp[j] * 16) = v;
                if (want_lo) {
                    f16x8 l;
#pragma unroll
                    for (int c = 0; c < 8; ++c) l[c] = c < nch ? (_Float16)((t[j][c] * iv[j] - (float)v[c]) * 2048.f) : (_Float16)0.f;
                    *reinterpret_cast<f16x8*>(xl + pp[j] * 16) = l;
                }
            }
        }
    }
    __syncthreads();
    char* blob0 = a.blob[o] + (size_t)n * nt * L.bytes;
    for (int id = tid; id < nt * 32; id += 256) {          // C part: granule g of every position
        const int tl = id >> 5, qq = id & 31;
        *reinterpret_cast<uint4*>(blob0 + (size_t)tl * L.bytes + L.c(qq, g)) = reinterpret_cast<const uint4*>(xt)[id];
        if (want_lo)
            *reinterpret_cast<uint4*>(a.clo[o] + ((size_t)n * nt + tl) * (KD * 64) + (g * 32 + qq) * 16) = reinterpret_cast<const uint4*>(xl)[id];
    }
    for (int id = tid; id < nt * 32; id += 256) {          // P part: granule cc of channel d = 8 g + c
        const int tl = id >> 5, cc = (id >> 3) & 3, c = id & 7;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pl = 16 * (cc >> 1) + 8 * ((e >> 2) & 1) + 4 * (cc & 1) + (e & 3);   // dg_perm32(pl) == 8 cc + e
            v[e] = *reinterpret_cast<const _Float16*>(xt + (tl * 32 + pl) * 16 + c * 2);
        }
        *reinterpret_cast<f16x8*>(blob0 + (size_t)tl * L.bytes + L.p(8 * g + c, cc)) = v;
    }
    for (int id = tid; id < nt * 8; id += 256) {           // per-tile column sums (for the cd means)
        const int tl = id >> 3, c = id & 7;
        float cs = 0.f;
#pragma unroll 8
        for (int qq = 0; qq < 32; ++qq) cs += (float)*reinterpret_cast<const _Float16*>(xt + (tl * 32 + qq) * 16 + c * 2);
        a.ccolpart[o][((size_t)n * nt + tl) * KD + 8 * g + c] = cs;
    }
}


// One launch prepares everything the fused kernel needs on the identity grid:
//   z = 0,1: feats operands (one block per tile), z = 2,3: code operands / norms, z = 4: depth indicators.
// grid tiles * B * (4 or 5) blocks (1-D, XCD-aware image-major order), block 256, dynamic LDS = max of the roles.
template <int MAXU, int UNC, bool FK = false>
__global__ __launch_bounds__(256) void k_prep_dense(const DgDenseArgs a) {
    extern __shared__ float sl[];
    // XCD-aware order (blocks are dealt round-robin over the 8 XCDs): consecutive logical ids - the source rows of one
    // image, which share 128-byte lines of the NCHW map - get one XCD, so a line is fetched into one L2 once
    int bid;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    }
    const int gx = a.Ppad / 32;                          // blocks per (image, role): one per tile (>= what the code roles need: launcher)
    const int nz = a.depth ? 5 : 4;                      // image-major: every XCD gets whole images, all roles
    const int x = bid % gx, z = (bid / gx) % nz, n = bid / (gx * nz);
    const int roles = a.roles ? a.roles : 15;
    if (n >= a.B) {                                      // the trailing blocks: the negatives' batch maps of this step
        if (roles & 8) dg_super_perm_row(nullptr, a.draw_seed, a.draw_state, a.B, a.draw_out, bid - gx * nz * a.B, a.draw_count, sl);
        return;
    }
    if (!(roles & (z < 2 ? 1 : (z < 4 ? 2 : 4)))) return;         // (a role of the other launch of a split call)
    if (z < 2) {
        if (!(DG_DBG(a.debug) & 1)) prep_dense_feats<MAXU, FK>(a, sl, x, n, z);
    } else if (z < 4) {
        if (a.code_split) {
            if (x * 256 < a.h * a.w) prep_dense_code_norms<UNC>(a, x, n, z - 2);
        } else if (x * DENSE_TPB < a.Ppad / 32 && !(DG_DBG(a.debug) & 2)) prep_dense_code<UNC>(a, sl, x, n, z - 2);
    } else if (x == 0 && !(DG_DBG(a.debug) & 4)) {
        depth_nz_image(a.depth, a.nz, a.nzsum, n, a.dH, a.dW, a.h, a.h, a.Ppad, true);
    }
}

// sum over the channels of sample(feats, coords)^2 per position (sample(): src/modules.py:822-825, the arithmetic of k_gather_norm
// above: align_corners, border clamp, output position (i, j) reads coords[n][j][i]) from the NCHW map directly - the norm-only pass of a
// call whose feature maps are wider than the operand kernels hold (dg_sampled_sumsq): out[n][p] (+)= the chunk's share.
// Block = 64 positions x 4 channel slices (slice q walks channels q, q + 4, ...; the four shares are added in slice order).
struct DgSumsqArgs { const float* feats; const float* coords; const int64_t* srcidx; float* out; int32_t B, C, h, w, S, Sh, P, accumulate; };
__global__ __launch_bounds__(256) void k_sampled_sumsq(const DgSumsqArgs a) {
    __shared__ float part[4][64];
    const int px = threadIdx.x & 63, q = threadIdx.x >> 6, p = blockIdx.x * 64 + px, n = blockIdx.y;
    const int pc = p < a.P ? p : a.P - 1, i = pc / a.S, j = pc - i * a.S;
    const float* c = a.coords + (((size_t)n * a.S + j) * a.Sh + i) * 2;
    float x = ((c[0] + 1.f) / 2.f) * (float)(a.w - 1);
    float y = ((c[1] + 1.f) / 2.f) * (float)(a.h - 1);
    x = fminf(fmaxf(x, 0.f), (float)(a.w - 1));
    y = fminf(fmaxf(y, 0.f), (float)(a.h - 1));
    const float x0f = floorf(x), y0f = floorf(y);
    const float wx1 = x - x0f, wy1 = y - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const bool inx = x0 + 1 <= a.w - 1, iny = y0 + 1 <= a.h - 1;
    const float w00 = wy0 * wx0, w01 = (inx ? wy0 * wx1 : 0.f), w10 = (iny ? wy1 * wx0 : 0.f), w11 = (inx && iny) ? wy1 * wx1 : 0.f;
    const int o00 = y0 * a.w + x0, o01 = o00 + (inx ? 1 : 0), o10 = o00 + (iny ? a.w : 0), o11 = o10 + (inx ? 1 : 0);
    const int ns = a.srcidx ? (int)a.srcidx[n] : n;
    const float* img = a.feats + (size_t)ns * a.C * a.h * a.w;
    const int HW = a.h * a.w;
    float ss = 0.f;
    for (int ch = q; ch < a.C; ch += 4) {
        const float* pl = img + (size_t)ch * HW;
        float v = pl[o00] * w00;
        if (w01 != 0.f) v += pl[o01] * w01;
        if (w10 != 0.f) v += pl[o10] * w10;
        if (w11 != 0.f) v += pl[o11] * w11;
        ss = fmaf(v, v, ss);
    }
    part[q][px] = ss;
    __syncthreads();
    if (q == 0 && p < a.P) {
        const float t = (part[0][px] + part[1][px]) + (part[2][px] + part[3][px]);
        float* o = a.out + (size_t)n * a.P + p;
        *o = a.accumulate ? *o + t : t;
    }
}
hipError_t dg_launch_sampled_sumsq(const float* feats, const float* coords, const int64_t* srcidx, float* out, int B, int C, int h, int w,
                                   int S, int Sh, int accumulate, hipStream_t s) {
    const DgSumsqArgs a{feats, coords, srcidx, out, B, C, h, w, S, Sh, S * Sh, accumulate};
    hipLaunchKernelGGL(k_sampled_sumsq, dim3((S * Sh + 63) / 64, B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// norm() over all C channels of an NCHW map, written as channel chunks (dg_normalize_split: the operands of DG_FEATS_UNIT calls).
// Block = 64 pixels x 4 channel slices (slice q sums channels q, q + 4, ... in order; the four partial sums are added in slice order).
struct DgNormSplitArgs { const float* src; float* dst[16]; int32_t B, C, P, nchunks, chunk_c; };
__global__ __launch_bounds__(256) void k_normalize_split(const DgNormSplitArgs a) {
    __shared__ float part[4][64];
    const int px = threadIdx.x & 63, q = threadIdx.x >> 6, p = blockIdx.x * 64 + px, b = blockIdx.y;
    const float* s = a.src + (size_t)b * a.C * a.P + (p < a.P ? p : a.P - 1);
    float ss = 0.f;
    for (int c0 = q; c0 < a.C; c0 += 32) {               // eight loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = c0 + 4 * u < a.C ? s[(size_t)(c0 + 4 * u) * a.P] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) ss = fmaf(v[u], v[u], ss);
    }
    part[q][px] = ss;
    __syncthreads();
    const float inv = 1.f / fmaxf(sqrtf((part[0][px] + part[1][px]) + (part[2][px] + part[3][px])), DG_EPS_NORM);
    if (p >= a.P) return;
    for (int c = q; c < a.C; c += 4) {
        const int k = c / a.chunk_c, ck = min(a.chunk_c, a.C - k * a.chunk_c);
        a.dst[k][((size_t)b * ck + (c - k * a.chunk_c)) * a.P + p] = s[(size_t)c * a.P] * inv;
    }
}
hipError_t dg_launch_normalize_split(const float* src, int B, int C, int P, int nchunks, int chunk_c, float* const* dst, hipStream_t s) {
    DgNormSplitArgs a = {};
    a.src = src; a.B = B; a.C = C; a.P = P; a.nchunks = nchunks; a.chunk_c = chunk_c;
    for (int k = 0; k < nchunks; ++k) a.dst[k] = dst[k];
    hipLaunchKernelGGL(k_normalize_split, dim3((P + 63) / 64, B), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t dg_launch_prep_dense(const DgDenseArgs& a, hipStream_t s) {
    if (a.KF > 768 || a.D > 128) return hipErrorInvalidValue;
    const int nt = a.Ppad / 32, gx = nt;
    if ((nt + DENSE_TPB - 1) / DENSE_TPB > gx || (a.h * a.w + 255) / 256 > gx) return hipErrorInvalidValue;
    const int smem = max(max(32 * (a.KF * 2 + 16) + 8 * 32 * 4,
                             DENSE_TPB * 32 * (a.KD + 4) * 2 + 4 * DENSE_CODE_PAIRS * 4 + DENSE_TPB * 32 * 4), a.draw_count > 0 ? a.B * 4 : 0);
    if (a.h * ((DENSE_TPB * 32 + a.h - 1) / a.h + 1) > DENSE_CODE_PAIRS) return hipErrorInvalidValue;   // pairs per block
    DgDenseArgs a2 = a;
#ifdef DG_DEVTOOLS
    if (const char* dbg = getenv("DG_PREP_DEBUG")) a2.debug = atoi(dbg);
#endif
    const dim3 grid(gx * a.B * (a.depth ? 5 : 4) + (a.draw_count > 0 ? a.draw_count : 0));
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a2);
        return hipGetLastError();
    };
    // registers per thread follow the widths
    if (a.fkeep[0] || a.fkeep[1]) {      // (deferred Dropout2d of the feature maps: dg_corr_forward_masked)
        if (a.D <= 72) return a.KF <= 384 ? launch(k_prep_dense<48, 18, true>) : launch(k_prep_dense<96, 18, true>);
        return a.KF <= 384 ? launch(k_prep_dense<48, 32, true>) : launch(k_prep_dense<96, 32, true>);
    }
    if (a.D <= 72) return a.KF <= 384 ? launch(k_prep_dense<48, 18>) : launch(k_prep_dense<96, 18>);
    return a.KF <= 384 ? launch(k_prep_dense<48, 32>) : launch(k_prep_dense<96, 32>);
}

// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------

// out[n][k] = scale * sum over the groups of part[n][group][k], summed in group order; every thread of the block takes part
__device__ __forceinline__ void colsum_reduce(const float* part, int n, int ngroups, int K, float scale, float* out) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float* cp = part + (size_t)n * ngroups * K + k;
        float s = 0.f;
        int t = 0;
        for (; t + 16 <= ngroups; t += 16) {          // 16 independent loads in flight, summed in group order
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = cp[(size_t)(t + u) * K];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; t + 4 <= ngroups; t += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = cp[(size_t)(t + u) * K];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; t < ngroups; ++t) s += cp[(size_t)t * K];
        out[(size_t)n * K + k] = s * scale;
    }
}

// bbar[o][n][k] = (1/P) sum over tiles of the per-tile column sums.  grid (B, nops), block 256.
__global__ __launch_bounds__(256) void k_colmean(const DgColmeanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char cm_smem[];     // blockIdx.z == 3: [Ppad] 16-byte code rows
    const int n = blockIdx.x, o = blockIdx.y;
    if (a.zero_word && n == 0 && o == 0 && blockIdx.z == 0 && threadIdx.x == 0) *a.zero_word = 0u;
    if (a.zero_words9 && n == 0 && o == 0 && blockIdx.z == 0 && threadIdx.x < 9) a.zero_words9[threadIdx.x] = 0u;
    // (always_inline: out of line, the lambda's pointers lose their address space - flat loads, tests/test_host_cpu.py audits it)
    auto reduce = [&](const float* part, int ngroups, int K, float scale, float* out) __attribute__((always_inline)) { colsum_reduce(part, n, ngroups, K, scale, out); };
    if (a.zsel == 1 ? blockIdx.z != 3 : (a.zsel == 2 && blockIdx.z == 3)) return;      // (one half of a split launch)
    if (blockIdx.z == 3) {                                // dense code operands from channel planes: y = operand * (KD / 8) + channel group
        const int GD = a.dc.KD / 8;
        if ((int)blockIdx.y < 2 * GD) dense_code_planes(a.dc, cm_smem, n, (int)blockIdx.y / GD, (int)blockIdx.y % GD);
        return;
    }
    // blockIdx.z: 0 = feature means, 1 = code column sums (two short latency chains side by side instead of one after the other),
    // 2 = consumer lists of k_corr2's grouped ragged blocks (one wave per (image, key))
    if (blockIdx.z == 2) {
        if ((int)blockIdx.y < a.gr.nkeys && threadIdx.x < 64) dg_group_lists(a.gr, n, o, threadIdx.x);
        return;
    }
    if (o >= a.nops) return;
    if (blockIdx.z == 0 && a.colpart[o]) {
        reduce(a.colpart[o], a.ngroups[o], a.KF, 1.f / (float)a.P, a.bbar[o]);
        // bbar as two bf16 halves (hi + lo keeps ~16 mantissa bits): the B fragments of k_rowmean, [n][2][KF]
        __syncthreads();
        for (int k = threadIdx.x; k < a.KF; k += 256) {
            const float v = a.bbar[o][(size_t)n * a.KF + k];
            const __bf16 hi = (__bf16)v;
            a.bsplit[o][((size_t)n * 2) * a.KF + k] = hi;
            a.bsplit[o][((size_t)n * 2 + 1) * a.KF + k] = (__bf16)(v - (float)hi);
        }
    }
    if (blockIdx.z == 1 && a.ccolpart[o] && a.dc.B == 0) reduce(a.ccolpart[o], a.Ppad / 32, a.KD, 1.f, a.csum[o]);
}

hipError_t dg_launch_colmean(const DgColmeanArgs& a, hipStream_t s) {
    int ny = a.gr.nkeys > a.nops ? a.gr.nkeys : a.nops;
    int nz = a.gr.nkeys > 0 ? 3 : 2, smem = 0;
    if (a.dc.B > 0) {
        if (a.nops != 2) return hipErrorInvalidValue;       // (the dense path has two operands; their csum moves to the k_rowmean launch)
        ny = max(ny, 2 * (a.dc.KD / 8)); nz = 4; smem = a.dc.Ppad * 16 * ((a.dc.clo[0] || a.dc.clo[1]) ? 2 : 1);
        hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_colmean), smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_colmean, dim3(a.B, ny, nz), dim3(256), smem, s, a);
    return hipGetLastError();
}

// r[t][n][p] = a[n][p] . bbar_t[n] for all pair-sets t at once: one wave per operand-1 tile (32 positions) runs a
// 32x32 MFMA chain over K with the tile's swizzled bf16 rows as A (read once for all pair-sets) and, as B, one column
// per pair-set holding bbar split into two bf16 halves (columns t and 16 + t: hi + lo keeps ~16 mantissa bits, the
// products with the bf16 rows are exact in the fp32 accumulator).  grid (Ppad/32, B), block 64.
#define ROWMEAN_WAVES 4
__global__ __launch_bounds__(64 * ROWMEAN_WAVES) void k_rowmean(const DgRowmeanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char rm_smem[];     // the tile's F part (32 bf16 rows, dg_f_off layout)
    const int tile = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KF = a.KF, nt = a.Ppad / 32;
    if (tile == nt + 1) {                                 // code column sums of the dense operands (DgDenseArgs.code_split)
        for (int o = 0; o < a.ncs; ++o) colsum_reduce(a.cs_part[o], n, nt, a.KD, 1.f, a.cs_out[o]);
        return;
    }
    if (tile == nt) {
        // extra block of the image: per-image sums of the row means (the fused kernel adds the B of them up to m0 = the
        // reference's fd.mean() before centering, src/modules.py:1237): sum_p a_p . bbar = P * abar . bbar, one K-long dot per
        // pair-set; the waves split the pair-sets (it used to trail the tile-0 wave as a serial chain of njobs dots)
        float av[12];                                   // KF <= 768: 12 values per lane
#pragma unroll
        for (int j = 0; j < 12; ++j) av[j] = lane + 64 * j < KF ? a.abar[(size_t)n * KF + lane + 64 * j] : 0.f;
        for (int t = wid; t < a.njobs; t += ROWMEAN_WAVES) {
            const DgRowmeanJob& Jt = a.jobs[t];
            const float* bt = Jt.bbar + (size_t)(Jt.bidx ? (int)Jt.bidx[n] : n) * KF;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) s = fmaf(av[j], lane + 64 * j < KF ? bt[lane + 64 * j] : 0.f, s);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) Jt.rimg[n] = s * (float)a.P;
        }
        return;
    }
    const DgBlob L(a.KF, a.KD);
    const int jb = c & 15, lo = c >> 4;
    const bool has = jb < a.njobs;
    const DgRowmeanJob& J = a.jobs[has ? jb : 0];
    const int nb = J.bidx ? (int)J.bidx[n] : n;
    // the four waves split K: wave w stages and multiplies the k-steps [w * nks, (w + 1) * nks) - a quarter of the F part
    // comes in linearly by LDS-DMA (1 KiB per instruction, fully coalesced); the B fragments (pre-split bbar) are plain
    // 16-byte loads.  (One wave per tile left the kernel latency-bound: three waves per CU.)
    const int nks = KF / 16 / ROWMEAN_WAVES;                         // 2, 6 or 12 (KF in {128, 384, 768})
    const int pieces = L.off_c / 1024 / ROWMEAN_WAVES;               // = nks: 16 k-values x 32 rows x 2 B = 1 KiB per k-step
    const char* fp = a.jobs[0].A + ((size_t)n * nt + tile) * L.bytes + lane * 16;
    const uint32_t dst = lds_addr(rm_smem);
    for (int pc = wid * pieces; pc < (wid + 1) * pieces; ++pc) dma16(fp + pc * 1024, dst + pc * 1024);
    const __bf16* bb = J.bsplit + ((size_t)nb * 2 + lo) * KF + 8 * h;
    f32x16 acc = {};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int ks0 = wid * nks; ks0 < (wid + 1) * nks; ks0 += 2) {     // nks is even
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ks = ks0 + u;
            af[u] = *reinterpret_cast<const bf16x8*>(rm_smem + dg_f_off(c, 2 * ks + h));      // A row q = c
            bf[u] = *reinterpret_cast<const bf16x8*>(bb + 16 * ks);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 b = bf[u];
            if (!has) {
#pragma unroll
                for (int e = 0; e < 8; ++e) b[e] = (__bf16)0.f;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u], b, acc, 0, 0, 0);
        }
    }
    // partial accumulators of waves 1..3 -> wave 0 (through the staging buffer, which every wave is done with), fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(rm_smem);
    if (wid > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((wid - 1) * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (wid != 0) return;
#pragma unroll
    for (int w = 0; w < ROWMEAN_WAVES - 1; ++w)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[(w * 16 + i) * 64 + lane];
    // lane (c, h) holds rows (i&3) + 8 (i>>2) + 4 h of column c; hi + lo columns are 16 lanes apart
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = acc[i] + __shfl(acc[i], (lane + 16) & 63, 64);
        const int p = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (has && !lo) J.rvec[(size_t)n * a.Ppad + p] = p < a.P ? v : 0.f;
        if (a.stash && has && !lo && jb == 0) {
            // k_corr2 FOLD (dg_corr2.hip): -r_p / 2 as hi + lo' / 2048 (hi = fp16(x), lo' = fp16(2048 (x - hi)): 22 bits, no subnormal
            // operand) in k = 0, 1 of position p's granule of the C part's first all-padding k-step; the other k stay zero
            const float x = p < a.P ? -0.5f * v : 0.f;
            const _Float16 xh = (_Float16)x, xl = (_Float16)((x - (float)xh) * 2048.f);
            const uint32_t word = (uint32_t)__builtin_bit_cast(unsigned short, xh) | ((uint32_t)__builtin_bit_cast(unsigned short, xl) << 16);
            *reinterpret_cast<uint32_t*>(a.stash + ((size_t)n * nt + tile) * L.bytes + L.off_c + a.stash_off + (p & 31) * 16) = word;
        }
    }
}
// the words above back to zero (dg_corr_materialize: k_corr_main's un-reduced forms multiply every code k-step of the blob) - or,
// with `rvec` (the intra pair-set's row means, still in the workspace), written again exactly as k_rowmean wrote them, so that the
// workspace is what the forward left (dg_corr_relaunch_main after a dg_corr_materialize)
__global__ __launch_bounds__(64) void k_set_stash(char* blobs, size_t blob_bytes, int off, const float* rvec, int ntiles, int P, int Ppad) {
    if (threadIdx.x >= 32) return;
    uint32_t word = 0u;
    if (rvec) {
        const int n = (int)blockIdx.x / ntiles, tile = (int)blockIdx.x - n * ntiles, p = tile * 32 + (int)threadIdx.x;
        const float x = p < P ? -0.5f * rvec[(size_t)n * Ppad + p] : 0.f;
        const _Float16 xh = (_Float16)x, xl = (_Float16)((x - (float)xh) * 2048.f);
        word = (uint32_t)__builtin_bit_cast(unsigned short, xh) | ((uint32_t)__builtin_bit_cast(unsigned short, xl) << 16);
    }
    *reinterpret_cast<uint32_t*>(blobs + (size_t)blockIdx.x * blob_bytes + off + threadIdx.x * 16) = word;
}
hipError_t dg_launch_set_stash(char* blobs, int B, int ntiles, size_t blob_bytes, int off, const float* rvec, int P, int Ppad, hipStream_t s) {
    hipLaunchKernelGGL(k_set_stash, dim3(B * ntiles), dim3(64), 0, s, blobs, blob_bytes, off, rvec, ntiles, P, Ppad);
    return hipGetLastError();
}

hipError_t dg_launch_rowmean(const DgRowmeanArgs& a, hipStream_t s) {
    if (a.njobs > 16) return hipErrorInvalidValue;
    const int smem = max(32 * (a.KF / 8) * 16, (ROWMEAN_WAVES - 1) * 16 * 64 * 4);     // F part / partial accumulators
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_rowmean), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_rowmean, dim3(a.Ppad / 32 + 1 + (a.ncs > 0 ? 1 : 0), a.B), dim3(64 * ROWMEAN_WAVES), smem, s, a);
    return hipGetLastError();
}

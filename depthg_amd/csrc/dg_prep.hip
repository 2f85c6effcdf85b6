// Operand preparation kernels (HBM-bound byte movers) for the correlation-loss path:
//   k_nchw_to_nhwc   (B,K,h,w) fp32 -> (B,h*w,K4) fp32, K4 = round_up(K,4), zero padded
//   k_gather_norm    sample() + norm() of the reference (src/modules.py:822-825, 789-790):
//                    bilinear gather at coords (grid_sample, border, align_corners=True),
//                    L2-normalise over channels, write the tile blobs (+ 1/norm, column sums)
//   k_depth_nz       depth -> F.interpolate(size=(S,S), bilinear, align_corners=True) -> norm over
//                    the single channel (src/modules.py:1261-1265): d / max(|d|, 1e-10)
//   k_rowmean        r[n][p] = a[n][p] . mean_q b[n][q]   (row means of fd for `pointwise`,
//                    src/modules.py:1236-1239 restated as a rank-1 term)
#include "dg_common.h"

// ------------------------------------------------------------------------------------------
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int K, int HW, int K4) {
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
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

hipError_t dg_launch_transpose(const float* src, float* dst, int B, int K, int HW, int K4, hipStream_t s) {
    dim3 grid((HW + 31) / 32, (K4 + 31) / 32, B), block(32, 8);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, block, 0, s, src, dst, K, HW, K4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------

// block = 256 threads = 4 waves; block handles 32 consecutive positions (one operand tile),
// wave w handles positions w, w+4, ...; lane l handles channels 4l + 256 m.  Output goes straight into
// the tile blob (dg_common.h): feats -> F part (bf16, swizzled rows), code -> C part (fp16, granule-major)
// and P part (fp16, P-major, dg_perm32 order; transposed through LDS).
template <int MAXM>
__global__ __launch_bounds__(256) void k_gather_norm(const DgGatherArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t ptile[128 * 32];   // [KD<=128][32] for the P part
    __shared__ float colred[4][MAXM * 256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int pt = blockIdx.x, n = blockIdx.y;
    const DgGatherJob& J = a.jobs[blockIdx.z];
    const int K4 = J.K4, Kpad = J.Kpad;
    const int ns = J.srcidx ? (int)J.srcidx[n] : n;
    const float* img = J.src + (size_t)ns * a.h * a.w * K4;
    const int S = a.S;
    const DgBlob L(a.KF, a.KD);
    char* blob = J.blob + ((size_t)n * (a.Ppad / 32) + pt) * L.bytes;

    float colacc[MAXM][4];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) colacc[m][0] = colacc[m][1] = colacc[m][2] = colacc[m][3] = 0.f;

    for (int pi = wid; pi < 32; pi += 4) {
        const int p = pt * 32 + pi;
        float4 v[MAXM];
#pragma unroll
        for (int m = 0; m < MAXM; ++m) v[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        float inv = 0.f;
        if (p < a.P) {
            // output position (i, j) = (p / S, p % S) reads x = coords[n][j][i][0], y = coords[n][j][i][1]
            const int i = p / S, j = p - i * S;
            const float* c = J.coords + (((size_t)n * S + j) * S + i) * 2;
            float x = ((c[0] + 1.f) / 2.f) * (float)(a.w - 1);
            float y = ((c[1] + 1.f) / 2.f) * (float)(a.h - 1);
            x = fminf(fmaxf(x, 0.f), (float)(a.w - 1));
            y = fminf(fmaxf(y, 0.f), (float)(a.h - 1));
            const float x0f = floorf(x), y0f = floorf(y);
            const float wx1 = x - x0f, wy1 = y - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            const int x0 = (int)x0f, y0 = (int)y0f;
            const bool inx = x0 + 1 <= a.w - 1, iny = y0 + 1 <= a.h - 1;
            const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
            const float* p00 = img + ((size_t)y0 * a.w + x0) * K4;
            const float* p01 = p00 + K4;
            const float* p10 = p00 + (size_t)a.w * K4;
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
        // normalise, convert, store (zero rows for p >= P, zero columns for k >= K4)
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            const int k = 4 * lane + 256 * m;
            if (k < Kpad) {
                float4 u = make_float4(v[m].x * inv, v[m].y * inv, v[m].z * inv, v[m].w * inv);
                colacc[m][0] += u.x; colacc[m][1] += u.y; colacc[m][2] += u.z; colacc[m][3] += u.w;
                uint2 o;
                if (J.is_code) {
                    f16x4 t; t[0] = (_Float16)u.x; t[1] = (_Float16)u.y; t[2] = (_Float16)u.z; t[3] = (_Float16)u.w;
                    o = *reinterpret_cast<uint2*>(&t);
                    *reinterpret_cast<uint2*>(blob + L.c(pi, k >> 3) + (k & 7) * 2) = o;
                    const int pp = dg_perm32(pi);
                    const uint16_t* ob = reinterpret_cast<const uint16_t*>(&o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ptile[(k + e) * 32 + pp] = ob[e];
                } else {
                    bf16x4 t; t[0] = (__bf16)u.x; t[1] = (__bf16)u.y; t[2] = (__bf16)u.z; t[3] = (__bf16)u.w;
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
            uint4 val = *reinterpret_cast<const uint4*>(&ptile[d * 32 + cc * 8]);
            *reinterpret_cast<uint4*>(blob + L.p(d, cc)) = val;
        }
    }
}

hipError_t dg_launch_gather(const DgGatherArgs& a, int maxK4, hipStream_t s) {
    dim3 grid(a.Ppad / 32, a.B, a.njobs), block(256);
    if (maxK4 <= 256)      hipLaunchKernelGGL(k_gather_norm<1>, grid, block, 0, s, a);
    else if (maxK4 <= 512) hipLaunchKernelGGL(k_gather_norm<2>, grid, block, 0, s, a);
    else if (maxK4 <= 768) hipLaunchKernelGGL(k_gather_norm<3>, grid, block, 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Dense identity grid (S == h == w, coords = the pixel centres): sample() is an exact spatial transpose
// (out[b,:,i,j] = t[b,:,j,i], reference quirk Q3), so the feats operand is built straight from the NCHW map: no
// channel-last copy, no bilinear taps.  One block per (source row y, image): reads the K x w slab of that row
// (w-float segments), normalises the w positions p = x*S + y and writes their swizzled bf16 rows into the tile blobs.
// grid (h, B, nops), block 256, dynamic LDS w * (KF + 1) floats.
struct DgDenseFeatsArgs {
    const float* src[2];     // NCHW fp32 (B,K,h,w)
    char* blob[2];
    float* colpart[2];       // [B][h][KF] per-source-row column sums
    int32_t B, K, KF, KD, h, w, P, Ppad;
};

__global__ __launch_bounds__(256) void k_prep_dense_feats(const DgDenseFeatsArgs a) {
    extern __shared__ float sl[];                  // [w][KF + 1]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int y = blockIdx.x, n = blockIdx.y, o = blockIdx.z;
    const int K = a.K, KF = a.KF, w = a.w, h = a.h, S = a.h, LD = KF + 1;
    const DgBlob L(a.KF, a.KD);
    const float* src = a.src[o] + (size_t)n * K * h * w + (size_t)y * w;
    // load: channel k, pixel x  (x fastest: w contiguous floats per channel)
    {
        constexpr int UN = 12;                      // independent loads in flight per thread
        const int x = tid & 31, k0 = tid >> 5;      // 8 channels per sweep of the block
        for (int kb = k0; kb < KF; kb += 8 * UN) {
            float t[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int k = kb + 8 * u;
                t[u] = (x < w && k < K) ? src[(size_t)k * h * w + x] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int k = kb + 8 * u;
                if (x < w && k < KF) sl[x * LD + k] = t[u];
            }
        }
    }
    __syncthreads();
    // one wave per position x: normalise over channels, write the blob row, accumulate column sums
    float colacc[3][4];
#pragma unroll
    for (int m = 0; m < 3; ++m) colacc[m][0] = colacc[m][1] = colacc[m][2] = colacc[m][3] = 0.f;
    for (int x = wid; x < w; x += 4) {
        const float* row = sl + x * LD;
        float v[3][4];
        float ss = 0.f;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * lane + 256 * m + e;
                v[m][e] = k < KF ? row[k] : 0.f;
                ss = fmaf(v[m][e], v[m][e], ss);
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float inv = 1.f / fmaxf(sqrtf(ss), DG_EPS_NORM);
        const int p = x * S + y;                        // sample() output position (i, j) = (x, y)
        char* blob = a.blob[o] + ((size_t)n * (a.Ppad / 32) + (p >> 5)) * L.bytes;
        const int q = p & 31;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int k = 4 * lane + 256 * m;
            if (k < KF) {
                bf16x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float u = v[m][e] * inv; colacc[m][e] += u; t[e] = (__bf16)u; }
                *reinterpret_cast<uint2*>(blob + L.f(q, k >> 3) + (k & 7) * 2) = *reinterpret_cast<uint2*>(&t);
            }
        }
    }
    // zero rows of the ragged last tile (positions P .. Ppad-1), once per image
    if (y == 0) {
        for (int idx = tid; idx < (a.Ppad - a.P) * (KF / 4); idx += 256) {
            const int p = a.P + idx / (KF / 4), k = (idx % (KF / 4)) * 4;
            char* blob = a.blob[o] + ((size_t)n * (a.Ppad / 32) + (p >> 5)) * L.bytes;
            *reinterpret_cast<uint2*>(blob + L.f(p & 31, k >> 3) + (k & 7) * 2) = make_uint2(0u, 0u);
        }
    }
    __syncthreads();
    float* colred = sl;                              // reuse: [4][KF]
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * lane + 256 * m + e;
            if (k < KF) colred[wid * KF + k] = colacc[m][e];
        }
    __syncthreads();
    for (int k = tid; k < KF; k += 256)
        a.colpart[o][((size_t)n * h + y) * KF + k] = colred[k] + colred[KF + k] + colred[2 * KF + k] + colred[3 * KF + k];
}

hipError_t dg_launch_prep_dense_feats(const float* f0, const float* f1, char* blob0, char* blob1, float* cp0, float* cp1,
                                      int B, int K, int KF, int KD, int h, int w, int P, int Ppad, hipStream_t s) {
    DgDenseFeatsArgs a;
    a.src[0] = f0; a.src[1] = f1; a.blob[0] = blob0; a.blob[1] = blob1; a.colpart[0] = cp0; a.colpart[1] = cp1;
    a.B = B; a.K = K; a.KF = KF; a.KD = KD; a.h = h; a.w = w; a.P = P; a.Ppad = Ppad;
    if (w > 32 || KF > 768) return hipErrorInvalidValue;
    const int smem = max(w * (KF + 1), 4 * KF) * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_prep_dense_feats), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_prep_dense_feats, dim3(h, B, 2), dim3(256), smem, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// depth (B,1,H,W) -> nz[n][p] over the S x S resize, p = i*S + j (row major)
__global__ void k_depth_nz(const float* __restrict__ depth, float* __restrict__ nz, int B, int H, int W, int S, int Ppad) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Ppad) return;
    float out = 0.f;
    if (p < S * S) {
        const int i = p / S, j = p - i * S;
        const float sy = S > 1 ? (float)(H - 1) / (float)(S - 1) : 0.f;
        const float sx = S > 1 ? (float)(W - 1) / (float)(S - 1) : 0.f;
        const float fy = sy * (float)i, fx = sx * (float)j;
        int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
        const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* d = depth + (size_t)n * H * W;
        const float top = d[(size_t)y0 * W + x0] * lx0 + d[(size_t)y0 * W + x1] * lx1;
        const float bot = d[(size_t)y1 * W + x0] * lx0 + d[(size_t)y1 * W + x1] * lx1;
        const float v = top * ly0 + bot * ly1;
        out = v / fmaxf(fabsf(v), DG_EPS_NORM);
    }
    nz[(size_t)n * Ppad + p] = out;
}

hipError_t dg_launch_depth_nz(const float* depth, float* nz, int B, int H, int W, int S, int Ppad, hipStream_t s) {
    dim3 grid((Ppad + 127) / 128, B), block(128);
    hipLaunchKernelGGL(k_depth_nz, grid, block, 0, s, depth, nz, B, H, W, S, Ppad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------

// bbar[o][n][k] = (1/P) sum over tiles of the per-tile column sums.  grid (B, nops), block 256.
__global__ __launch_bounds__(256) void k_colmean(const DgColmeanArgs a) {
    const int n = blockIdx.x, o = blockIdx.y, nt = a.ngroups[o];
    const float invP = 1.f / (float)a.P;
    for (int k = threadIdx.x; k < a.KF; k += 256) {
        float s = 0.f;
        for (int t = 0; t < nt; ++t) s += a.colpart[o][((size_t)n * nt + t) * a.KF + k];
        a.bbar[o][(size_t)n * a.KF + k] = s * invP;
    }
}

hipError_t dg_launch_colmean(const DgColmeanArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_colmean, dim3(a.B, a.nops), dim3(256), 0, s, a);
    return hipGetLastError();
}

// r[n][p] = a[n][p] . bbar[n]; grid (nchunk, B, njobs), block 256: DG_RM_ROWS rows per block, one row per
// wave step; lanes walk the 16-byte granule slots of the swizzled F rows.
__global__ __launch_bounds__(256) void k_rowmean(const DgRowmeanArgs a) {
    __shared__ float bbar[768];
    __shared__ float wsum[4];
    const int ch = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const DgRowmeanJob& J = a.jobs[blockIdx.z];
    const int KF = a.KF, GF = KF / 8, nt = a.Ppad / 32;
    const DgBlob L(a.KF, a.KD);
    const int na = J.aidx ? (int)J.aidx[n] : n;
    const int nb = J.bidx ? (int)J.bidx[n] : n;
    for (int k = tid; k < KF; k += 256) bbar[k] = J.bbar[(size_t)nb * KF + k];
    __syncthreads();
    float tot = 0.f;
    for (int pi = wid; pi < DG_RM_ROWS; pi += 4) {
        const int p = ch * DG_RM_ROWS + pi;
        if (p >= a.Ppad) break;
        float d = 0.f;
        if (p < a.P) {
            const int q = p & 31;
            const char* row = J.A + ((size_t)na * nt + (p >> 5)) * L.bytes + (size_t)q * GF * 16;
            for (int slot = lane; slot < GF; slot += 64) {
                const int g = slot ^ (q & 15);            // slot holds granule g (involution)
                uint4 raw = *reinterpret_cast<const uint4*>(row + slot * 16);
                const uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d = fmaf(__uint_as_float(wds[e] << 16), bbar[8 * g + 2 * e], d);
                    d = fmaf(__uint_as_float(wds[e] & 0xffff0000u), bbar[8 * g + 2 * e + 1], d);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        }
        if (lane == 0) { J.rvec[(size_t)n * a.Ppad + p] = d; tot += d; }
    }
    if (lane == 0) wsum[wid] = tot;
    __syncthreads();
    if (tid == 0) J.rsum[(size_t)n * a.nchunk + ch] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// m0[t] = mean over valid (n, p) of rvec = the reference's fd.mean() before centering (fixed summation order)
__global__ __launch_bounds__(256) void k_m0(const DgRowmeanArgs a) {
    __shared__ float wsum[4];
    const DgRowmeanJob& J = a.jobs[blockIdx.x];
    const int tid = threadIdx.x, n = a.B * a.nchunk;
    float s = 0.f;
    for (int i = tid; i < n; i += 256) s += J.rsum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) a.m0[blockIdx.x][0] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) / ((float)a.B * (float)a.P);
}

hipError_t dg_launch_rowmean(const DgRowmeanArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_rowmean, dim3(a.nchunk, a.B, a.njobs), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_m0, dim3(a.njobs), dim3(256), 0, s, a);
    return hipGetLastError();
}

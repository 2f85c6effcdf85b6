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
    if (blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) { a.tickets[n] = 0; if (n == 0) a.tickets[a.B] = 0; }
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
__device__ __forceinline__ void prep_dense_feats(const DgDenseArgs& a, float* sl, int y, int n, int o) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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

// Code operand on the identity grid: one block per tile of 32 positions p = i*S + j <- pixel (y = j, x = i) of the NCHW
// code map; L2-normalise over the D channels (norm(), src/modules.py:789-790), write the C part (K-major granules), the
// P part (position-major granules in dg_perm32 order) and 1/max(||c||, eps).  Same roundings as k_gather_norm.
__device__ __forceinline__ void prep_dense_code(const DgDenseArgs& a, float* sl, int pt, int n, int o) {
    const int tid = threadIdx.x, q = tid & 31, kk = tid >> 5;
    const int KD = a.KD, D = a.D, S = a.h, HW = a.h * a.w, LD = KD + 1;
    float* xs = sl;                      // [32][KD + 1]
    float* red = sl + 32 * LD;           // [8][32] partial sums of squares, then [32] 1/norm at red[256..]
    const DgBlob L(a.KF, a.KD);
    const int p = pt * 32 + q;
    const bool valid = p < a.P;
    const int i = p / S, j = p - i * S;
    const float* src = a.code[o] + (size_t)n * D * HW + (valid ? j * a.w + i : 0);
    {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int k = kk + 8 * u; t[u] = (valid && k < D) ? src[(size_t)k * HW] : 0.f; }
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int k = kk + 8 * u; if (k < KD) { xs[q * LD + k] = t[u]; ss = fmaf(t[u], t[u], ss); } }
        red[kk * 32 + q] = ss;
    }
    __syncthreads();
    if (tid < 32) {
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) ss += red[u * 32 + tid];
        const float inv = (pt * 32 + tid) < a.P ? 1.f / fmaxf(sqrtf(ss), DG_EPS_NORM) : 0.f;
        red[256 + tid] = inv;
        a.inv_norm[o][(size_t)n * a.Ppad + pt * 32 + tid] = inv;
    }
    __syncthreads();
    char* blob = a.blob[o] + ((size_t)n * (a.Ppad / 32) + pt) * L.bytes;
    for (int id = tid; id < (KD / 8) * 32; id += 256) {       // C part: granule g of position qq
        const int g = id >> 5, qq = id & 31;
        const float inv = red[256 + qq];
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)(xs[qq * LD + 8 * g + e] * inv);
        *reinterpret_cast<f16x8*>(blob + L.c(qq, g)) = v;
    }
    for (int d = tid; d < KD; d += 256) {                     // per-tile column sums (for the cd means)
        float cs = 0.f;
#pragma unroll 8
        for (int qq = 0; qq < 32; ++qq) cs += xs[qq * LD + d] * red[256 + qq];
        a.ccolpart[o][((size_t)n * (a.Ppad / 32) + pt) * KD + d] = cs;
    }
    for (int id = tid; id < 4 * KD; id += 256) {              // P part: granule cc of channel d = slots 8cc .. 8cc+7
        const int cc = id / KD, d = id - cc * KD;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pl = 16 * (cc >> 1) + 8 * ((e >> 2) & 1) + 4 * (cc & 1) + (e & 3);   // dg_perm32(pl) == 8 cc + e
            v[e] = (_Float16)(xs[pl * LD + d] * red[256 + pl]);
        }
        *reinterpret_cast<f16x8*>(blob + L.p(d, cc)) = v;
    }
}

__device__ __forceinline__ void depth_nz_image(const float* __restrict__ depth, float* __restrict__ nz, float* __restrict__ nzsum,
                                               int n, int H, int W, int S, int Ppad);

// One launch prepares everything the fused kernel needs on the identity grid:
//   z = 0,1: feats operands (one block per source row), z = 2,3: code operands (one block per tile), z = 4: depth indicators.
// grid (max(h, Ppad/32), B, 4 or 5), block 256, dynamic LDS max(w*(KF+1), 4*KF, 32*(KD+1) + 288) floats.
__global__ __launch_bounds__(256) void k_prep_dense(const DgDenseArgs a) {
    extern __shared__ float sl[];
    const int z = blockIdx.z, n = blockIdx.y;
    if (z < 2) {
        if ((int)blockIdx.x < a.h) prep_dense_feats(a, sl, blockIdx.x, n, z);
    } else if (z < 4) {
        if ((int)blockIdx.x < a.Ppad / 32) prep_dense_code(a, sl, blockIdx.x, n, z - 2);
    } else if (blockIdx.x == 0) {
        depth_nz_image(a.depth, a.nz, a.nzsum, n, a.dH, a.dW, a.h, a.Ppad);
    }
    if (blockIdx.x == 0 && z == 0 && threadIdx.x == 0) { a.tickets[n] = 0; if (n == 0) a.tickets[a.B] = 0; }
}

hipError_t dg_launch_prep_dense(const DgDenseArgs& a, hipStream_t s) {
    if (a.w > 32 || a.KF > 768 || a.D > 128) return hipErrorInvalidValue;
    const int nt = a.Ppad / 32, gx = max(a.h, nt);
    const int smem = max(max(a.w * (a.KF + 1), 4 * a.KF), 32 * (a.KD + 1) + 288) * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_prep_dense), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_prep_dense, dim3(gx, a.B, a.depth ? 5 : 4), dim3(256), smem, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// depth (B,1,H,W) -> nz[n][p] over the S x S resize, p = i*S + j (row major)
__device__ __forceinline__ float depth_nz_at(const float* __restrict__ depth, int n, int p, int H, int W, int S) {
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
    return out;
}

// all positions of image n by one block of 256 threads, plus their sum (mean(dd) = mean_n (sum_p nz)^2 / P^2)
__device__ __forceinline__ void depth_nz_image(const float* __restrict__ depth, float* __restrict__ nz, float* __restrict__ nzsum,
                                               int n, int H, int W, int S, int Ppad) {
    __shared__ float wred[4];
    float s = 0.f;
    for (int p = threadIdx.x; p < Ppad; p += 256) {
        const float v = depth_nz_at(depth, n, p, H, W, S);
        nz[(size_t)n * Ppad + p] = v;
        s += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) nzsum[n] = wred[0] + wred[1] + wred[2] + wred[3];
}

__global__ __launch_bounds__(256) void k_depth_nz(const float* __restrict__ depth, float* __restrict__ nz, float* __restrict__ nzsum,
                                                  int H, int W, int S, int Ppad) {
    depth_nz_image(depth, nz, nzsum, blockIdx.x, H, W, S, Ppad);
}

hipError_t dg_launch_depth_nz(const float* depth, float* nz, float* nzsum, int B, int H, int W, int S, int Ppad, hipStream_t s) {
    hipLaunchKernelGGL(k_depth_nz, dim3(B), dim3(256), 0, s, depth, nz, nzsum, H, W, S, Ppad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------

// bbar[o][n][k] = (1/P) sum over tiles of the per-tile column sums.  grid (B, nops), block 256.
__global__ __launch_bounds__(256) void k_colmean(const DgColmeanArgs a) {
    const int n = blockIdx.x, o = blockIdx.y;
    auto reduce = [&](const float* part, int ngroups, int K, float scale, float* out) {
        for (int k = threadIdx.x; k < K; k += 256) {
            const float* cp = part + (size_t)n * ngroups * K + k;
            float s = 0.f;
            int t = 0;
            for (; t + 8 <= ngroups; t += 8) {            // 8 independent loads in flight, summed in group order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = cp[(size_t)(t + u) * K];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; t < ngroups; ++t) s += cp[(size_t)t * K];
            out[(size_t)n * K + k] = s * scale;
        }
    };
    if (a.colpart[o]) reduce(a.colpart[o], a.ngroups[o], a.KF, 1.f / (float)a.P, a.bbar[o]);
    if (a.ccolpart[o]) reduce(a.ccolpart[o], a.Ppad / 32, a.KD, 1.f, a.csum[o]);
}

hipError_t dg_launch_colmean(const DgColmeanArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_colmean, dim3(a.B, a.nops), dim3(256), 0, s, a);
    return hipGetLastError();
}

// r[t][n][p] = a[n][p] . bbar_t[n] for all pair-sets t at once: one wave per operand-1 tile (32 positions) runs a
// 32x32 MFMA chain over K with the tile's swizzled bf16 rows as A (read once for all pair-sets) and, as B, one column
// per pair-set holding bbar split into two bf16 halves (columns t and 16 + t: hi + lo keeps ~16 mantissa bits, the
// products with the bf16 rows are exact in the fp32 accumulator).  grid (Ppad/32, B), block 64.
__global__ __launch_bounds__(64) void k_rowmean(const DgRowmeanArgs a) {
    const int tile = blockIdx.x, n = blockIdx.y, lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int KF = a.KF, GF = KF / 8, nt = a.Ppad / 32;
    const DgBlob L(a.KF, a.KD);
    const int jb = c & 15, lo = c >> 4;
    const bool has = jb < a.njobs;
    const DgRowmeanJob& J = a.jobs[has ? jb : 0];
    const int nb = J.bidx ? (int)J.bidx[n] : n;
    const float* bb = J.bbar + (size_t)nb * KF + 8 * h;
    const char* row = a.jobs[0].A + ((size_t)n * nt + tile) * L.bytes + (size_t)c * GF * 16;   // A row q = c
    f32x16 acc = {};
    for (int ks0 = 0; ks0 < KF / 16; ks0 += 8) {     // KF / 16 is a multiple of 8 (KF in {128, 384, 768}); 24 loads in flight
        bf16x8 af[8];
        float4 b0[8], b1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ks = ks0 + u;
            af[u] = *reinterpret_cast<const bf16x8*>(row + (((2 * ks + h) ^ (c & 15)) * 16));
            b0[u] = *reinterpret_cast<const float4*>(bb + 16 * ks);
            b1[u] = *reinterpret_cast<const float4*>(bb + 16 * ks + 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float bv[8] = {b0[u].x, b0[u].y, b0[u].z, b0[u].w, b1[u].x, b1[u].y, b1[u].z, b1[u].w};
            bf16x8 bf;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 hi = (__bf16)bv[e];
                bf[e] = !has ? (__bf16)0.f : (lo ? (__bf16)(bv[e] - (float)hi) : hi);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u], bf, acc, 0, 0, 0);
        }
    }
    // lane (c, h) holds rows (i&3) + 8 (i>>2) + 4 h of column c; hi + lo columns are 16 lanes apart
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = acc[i] + __shfl(acc[i], (lane + 16) & 63, 64);
        const int p = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const float d = p < a.P ? v : 0.f;
        if (has && !lo) J.rvec[(size_t)n * a.Ppad + p] = d;
        tot += d;
    }
    tot += __shfl_xor(tot, 32, 64);
    // per-image sums of the row means (the fused kernel adds the B of them up to m0 = the reference's fd.mean() before
    // centering, src/modules.py:1237): per-tile sums are published, the last wave of the image (ticket zeroed by the
    // operand-preparation kernel; nt waves per ticket) adds them in tile order
    if (has && !lo && h == 0) dg_publish(J.rtile + (size_t)n * nt + tile, tot);
    int last = 0;
    if (lane == 0) last = atomicAdd(a.tickets + n, 1) == nt - 1;
    last = __shfl(last, 0, 64);
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (has && !lo && h == 0) {
        float s = 0.f;
        for (int t = 0; t < nt; ++t) s += dg_read_published(J.rtile + (size_t)n * nt + t);
        J.rimg[n] = s;
    }
    if (lane == 0) atomicExch(a.tickets + n, 0);
}

hipError_t dg_launch_rowmean(const DgRowmeanArgs& a, hipStream_t s) {
    if (a.njobs > 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_rowmean, dim3(a.Ppad / 32, a.B), dim3(64), 0, s, a);
    return hipGetLastError();
}

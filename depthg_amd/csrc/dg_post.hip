// Backward tail and depth-guided sampling kernels:
//   k_scatter_grad  grid_sample backward (bilinear, border, align_corners=True; the adjoint of
//                   sample(), reference src/modules.py:822-825) of the per-pair-set code gradients,
//                   combined with the upstream gradients of the four loss means, accumulated per
//                   destination image in LDS (no global atomics), written as (B,D,h,w) fp32.
//   k_fps_coords    farthest_point_sampling_depth (src/modules.py:999-1037) = adaptive_avg_pool2d
//                   -> depth2points(fov=90 rad, :988-996) -> fps (:939-985) -> row-major coords*2-1.
#include "dg_common.h"


__device__ __forceinline__ void dg_taps(const float* c, int h, int w, int& x0, int& y0, bool& inx, bool& iny,
                                        float& w00, float& w01, float& w10, float& w11) {
    float x = ((c[0] + 1.f) / 2.f) * (float)(w - 1);
    float y = ((c[1] + 1.f) / 2.f) * (float)(h - 1);
    x = fminf(fmaxf(x, 0.f), (float)(w - 1));
    y = fminf(fmaxf(y, 0.f), (float)(h - 1));
    const float x0f = floorf(x), y0f = floorf(y);
    const float wx1 = x - x0f, wy1 = y - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    x0 = (int)x0f; y0 = (int)y0f;
    inx = x0 + 1 <= w - 1; iny = y0 + 1 <= h - 1;
    w00 = wy0 * wx0; w01 = wy0 * wx1; w10 = wy1 * wx0; w11 = wy1 * wx1;
}

// Stage 1: comb[dest][b][p][:] = sum over the direct sources of `dest` of factor * upstream * buf[b][p][:]
// (elementwise, float4).  grid (ceil(B*Ppad*DP/4 / 256), 2).
__global__ __launch_bounds__(256) void k_grad_combine(const DgScatterArgs a) {
    const int dest = blockIdx.y;
    const size_t n4 = (size_t)a.B * a.Ppad * a.DP / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < a.nsrc; ++s) {
        const DgScatterSrc& q = a.src[s];
        if (q.dest != dest || q.route != nullptr) continue;
        const float sc = q.factor * a.gscal[q.gidx];
        const float4 v = reinterpret_cast<const float4*>(q.buf)[i];
        acc.x = fmaf(sc, v.x, acc.x); acc.y = fmaf(sc, v.y, acc.y); acc.z = fmaf(sc, v.z, acc.z); acc.w = fmaf(sc, v.w, acc.w);
    }
    reinterpret_cast<float4*>(a.comb[dest])[i] = acc;
}

#define SCAT_THREADS 512
__device__ __forceinline__ void scatter_rows(float* acc, int stride, const float* buf, float sc, const float* coords,
                                             int nimg, int d, int dl, int pl, int pstep, const DgScatterArgs& a) {
    const int S = a.S;
    for (int p = pl; p < a.P; p += pstep) {
        const int i = p / S, j = p - i * S;
        int x0, y0; bool inx, iny; float w00, w01, w10, w11;
        dg_taps(coords + (((size_t)nimg * S + j) * S + i) * 2, a.h, a.w, x0, y0, inx, iny, w00, w01, w10, w11);
        if (d < a.D) {
            const float v = sc * buf[((size_t)nimg * a.Ppad + p) * a.DP + d];
            float* base = acc + dl * stride + y0 * a.w + x0;
            if (w00 != 0.f) __hip_atomic_fetch_add(base, v * w00, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (inx && w01 != 0.f) __hip_atomic_fetch_add(base + 1, v * w01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (iny && w10 != 0.f) __hip_atomic_fetch_add(base + a.w, v * w10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (inx && iny && w11 != 0.f) __hip_atomic_fetch_add(base + a.w + 1, v * w11, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// Stage 2: grid (ceil(D/DC), B, 2), block SCAT_THREADS, dynamic LDS DC*(h*w+1)*4: one destination image and
// channel chunk per block, accumulated in LDS, written once.
__global__ __launch_bounds__(SCAT_THREADS) void k_scatter_grad(const DgScatterArgs a) {
    extern __shared__ float acc[];
    const int tid = threadIdx.x;
    const int dc0 = blockIdx.x * a.DC, bdst = blockIdx.y, dest = blockIdx.z;
    const int HW = a.h * a.w, stride = HW + 1;
    for (int i = tid; i < a.DC * stride; i += SCAT_THREADS) acc[i] = 0.f;
    __syncthreads();
    const int dl = tid & (a.DC - 1), pl = tid / a.DC, pstep = SCAT_THREADS / a.DC;
    const int d = dc0 + dl;
    // direct sources (already combined): image bdst -> destination bdst; coords1 for grad_code, coords2 for grad_code_pos
    scatter_rows(acc, stride, a.comb[dest], 1.0f, dest == 0 ? a.coords1 : a.coords2, bdst, d, dl, pl, pstep, a);
    // routed sources (negatives): image n scatters into destination route[n] with n's coords
    for (int s = 0; s < a.nsrc; ++s) {
        const DgScatterSrc& q = a.src[s];
        if (q.dest != dest || q.route == nullptr) continue;
        const float sc = q.factor * a.gscal[q.gidx];
        const float* coords = q.coords_sel == 0 ? a.coords1 : a.coords2;
        for (int n = 0; n < a.B; ++n)
            if ((int)q.route[n] == bdst) scatter_rows(acc, stride, q.buf, sc, coords, n, d, dl, pl, pstep, a);
    }
    __syncthreads();
    float* out = a.out[dest];
    for (int i = tid; i < a.DC * HW; i += SCAT_THREADS) {
        const int dd = i / HW, pix = i - dd * HW;
        if (dc0 + dd < a.D) out[((size_t)bdst * a.D + dc0 + dd) * HW + pix] = acc[dd * stride + pix];
    }
}

hipError_t dg_launch_scatter(const DgScatterArgs& a, hipStream_t s) {
    const size_t n4 = (size_t)a.B * a.Ppad * a.DP / 4;
    hipLaunchKernelGGL(k_grad_combine, dim3((unsigned)((n4 + 255) / 256), 2), dim3(256), 0, s, a);
    const int smem = a.DC * (a.h * a.w + 1) * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter_grad), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    dim3 grid((a.D + a.DC - 1) / a.DC, a.B, 2);
    hipLaunchKernelGGL(k_scatter_grad, grid, dim3(SCAT_THREADS), smem, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// FPS.  One block per image.  All arithmetic is IEEE fp32 with the reference's operation order and
// no fused multiply-add, so that the selected set is bit-identical to numpy's for the same depth.
#define FPS_THREADS 256
__global__ __launch_bounds__(FPS_THREADS) void k_fps_coords(const float* __restrict__ depth, int H, int W, int h, int w,
                                                            int S, float factor, float* __restrict__ out_coords,
                                                            int32_t* __restrict__ out_inds) {
    extern __shared__ float sm[];
    const int HW = h * w, nsel = S * S;
    float* px = sm; float* py = px + HW; float* pz = py + HW; float* dist = pz + HW;
    int* sel = reinterpret_cast<int*>(dist + HW);          // 1 when selected
    __shared__ float rv[FPS_THREADS / 64];
    __shared__ int ri[FPS_THREADS / 64];
    __shared__ int s_last;
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* d = depth + (size_t)n * H * W;
    // adaptive_avg_pool2d + depth2points
    for (int idx = tid; idx < HW; idx += FPS_THREADS) {
        const int i = idx / w, j = idx - i * w;
        const int ys = (i * H) / h, ye = ((i + 1) * H + h - 1) / h;
        const int xs = (j * W) / w, xe = ((j + 1) * W + w - 1) / w;
        float s = 0.f;
        for (int y = ys; y < ye; ++y)
            for (int x = xs; x < xe; ++x) s = __fadd_rn(s, d[(size_t)y * W + x]);
        const float dv = __fdiv_rn(s, (float)((ye - ys) * (xe - xs)));
        const float fd = __fmul_rn(factor, dv);
        py[idx] = __fdiv_rn(__fmul_rn(fd, __fsub_rn((float)i, (float)h / 2.0f)), (float)h);
        px[idx] = __fdiv_rn(__fmul_rn(fd, __fsub_rn((float)j, (float)w / 2.0f)), (float)w);
        pz[idx] = __fmul_rn(-dv, 5.0f);
        dist[idx] = __builtin_inff();
        sel[idx] = 0;
    }
    if (tid == 0) { s_last = 0; if (out_inds) out_inds[(size_t)n * nsel] = 0; }
    __syncthreads();
    if (tid == 0) sel[0] = 1;
    __syncthreads();
    for (int it = 1; it < nsel; ++it) {
        const int last = s_last;
        const float lx = px[last], ly = py[last], lz = pz[last];
        float bv = -1.f; int bi = 0x7fffffff;
        for (int idx = tid; idx < HW; idx += FPS_THREADS) {
            if (sel[idx]) continue;
            const float dx = __fsub_rn(lx, px[idx]), dy = __fsub_rn(ly, py[idx]), dz = __fsub_rn(lz, pz[idx]);
            const float dd = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            const float nd = fminf(dd, dist[idx]);
            dist[idx] = nd;
            if (nd > bv) { bv = nd; bi = idx; }      // ascending idx per thread -> first max kept
        }
        // block argmax with lowest-index tie-break
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { rv[tid >> 6] = bv; ri[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = rv[0]; int i0 = ri[0];
            for (int k = 1; k < FPS_THREADS / 64; ++k)
                if (rv[k] > v || (rv[k] == v && ri[k] < i0)) { v = rv[k]; i0 = ri[k]; }
            s_last = i0; sel[i0] = 1;
            if (out_inds) out_inds[(size_t)n * nsel + it] = i0;
        }
        __syncthreads();
    }
    // selected set in row-major order -> coords (row/h, col/w)*2-1
    for (int idx = tid; idx < HW; idx += FPS_THREADS) {
        if (!sel[idx]) continue;
        int rank = 0;
        for (int k = 0; k < idx; ++k) rank += sel[k];
        const int i = idx / w, j = idx - i * w;
        float* o = out_coords + ((size_t)n * nsel + rank) * 2;
        o[0] = __fsub_rn(__fmul_rn(__fdiv_rn((float)i, (float)h), 2.0f), 1.0f);
        o[1] = __fsub_rn(__fmul_rn(__fdiv_rn((float)j, (float)w), 2.0f), 1.0f);
    }
}

hipError_t dg_launch_fps(const float* depth, int B, int H, int W, int h, int w, int S, float factor,
                         float* out_coords, int32_t* out_inds, hipStream_t s) {
    const int smem = h * w * 5 * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fps_coords), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fps_coords, dim3(B), dim3(FPS_THREADS), smem, s, depth, H, W, h, w, S, factor, out_coords, out_inds);
    return hipGetLastError();
}

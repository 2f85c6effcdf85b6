// The probes the caller trains next to the correlation loss, as gfx950 kernels:
//   * ClusterLookup (src/modules.py:647-675): cosine similarities of the normalised code with the normalised cluster centres, hard
//     (alpha None: one-hot of the arg-max) or soft (softmax(alpha * inner)) assignment, loss = -mean_{b,p} sum_n probs * inner;
//   * the linear probe's loss (src/train_segmentation.py:421-434): the probe's logits at feature resolution, bilinearly resized to
//     the label resolution (align_corners=False), cross entropy over the labelled pixels (0 <= label < n_classes), mean.
// The reference materialises (B*H*W, n_classes) logits at label resolution and three masks; here a label pixel's logits are blended
// in registers and the adjoint of the resize is a gather per low-resolution row (fixed order: no floating-point atomics).
#include "dg_common.h"

#define DG_NORM_EPS 1e-12f       // F.normalize default eps (src/modules.py:668-669 call it without one)


// one thread per position; the normalised centres sit in LDS ([n][D + 1])
__global__ __launch_bounds__(256) void k_cluster_fwd(const DgClusterArgs a) {
    extern __shared__ float csm[];
    float* nc = csm;                                  // [n][D + 1]
    __shared__ float red[4];
    const int D = a.D, n = a.n, P = a.P, ld = D + 1;
    for (int c = threadIdx.x; c < n; c += 256) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) { const float v = a.clusters[(size_t)c * D + d]; s = fmaf(v, v, s); }
        const float inv = 1.f / fmaxf(sqrtf(s), DG_NORM_EPS);
        for (int d = 0; d < D; ++d) nc[c * ld + d] = a.clusters[(size_t)c * D + d] * inv;
    }
    __syncthreads();
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    float contrib = 0.f;
    if (p < P) {
        const float* x = a.x + (size_t)b * D * P + p;
        float s = 0.f;
        for (int d = 0; d < D; ++d) { const float v = x[(size_t)d * P]; s = fmaf(v, v, s); }
        const float inv = 1.f / fmaxf(sqrtf(s), DG_NORM_EPS);
        const bool hard = a.alpha != a.alpha;
        float best = -INFINITY, mx = -INFINITY;
        int arg = 0;
        float* in = a.inner + (size_t)b * n * P + p;
        for (int c = 0; c < n; ++c) {
            float ip = 0.f;
            for (int d = 0; d < D; ++d) ip = fmaf(x[(size_t)d * P] * inv, nc[c * ld + d], ip);
            in[(size_t)c * P] = ip;
            if (ip > best) { best = ip; arg = c; }           // first maximum wins, as torch.argmax
            mx = fmaxf(mx, ip * a.alpha);
        }
        if (hard) {
            contrib = best;
            if (a.probs) for (int c = 0; c < n; ++c) a.probs[((size_t)b * n + c) * P + p] = c == arg ? 1.f : 0.f;
        } else {
            float z = 0.f;
            for (int c = 0; c < n; ++c) z += expf(in[(size_t)c * P] * a.alpha - mx);
            const float lz = logf(z);
            for (int c = 0; c < n; ++c) {
                const float ip = in[(size_t)c * P], lp = ip * a.alpha - mx - lz, pr = expf(lp);
                contrib = fmaf(pr, ip, contrib);
                if (a.probs) a.probs[((size_t)b * n + c) * P + p] = pr;
                if (a.logp) a.logp[((size_t)b * n + c) * P + p] = lp;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) a.part[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = scale * sum(part[0..n))   (one wave, fixed order)
__global__ __launch_bounds__(64) void k_sum_scale(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[0] = s * scale;
}

hipError_t dg_launch_cluster_fwd(const DgClusterArgs& a, float* loss_out, hipStream_t s) {
    const int nbx = (a.P + 255) / 256;
    const int smem = a.n * (a.D + 1) * 4;
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_cluster_fwd), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cluster_fwd, dim3(nbx, a.B), dim3(256), smem, s, a);
    hipLaunchKernelGGL(k_sum_scale, dim3(1), dim3(64), 0, s, a.part, nbx * a.B, -1.0f / ((float)a.B * (float)a.P), loss_out);
    return hipGetLastError();
}


// per position: d loss / d inner (hard: -g/(BP) at the arg-max; soft: -g/(BP) p_j (1 + alpha (i_j - sum_k p_k i_k))), and, if asked
// for, the gradient w.r.t. x through the normalisation
__global__ __launch_bounds__(256) void k_cluster_bwd_pos(const DgClusterBwdArgs a) {
    extern __shared__ float csm[];
    float* nc = csm;
    const int D = a.D, n = a.n, P = a.P, ld = D + 1;
    for (int c = threadIdx.x; c < n; c += 256) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) { const float v = a.clusters[(size_t)c * D + d]; s = fmaf(v, v, s); }
        const float inv = 1.f / fmaxf(sqrtf(s), DG_NORM_EPS);
        for (int d = 0; d < D; ++d) nc[c * ld + d] = a.clusters[(size_t)c * D + d] * inv;
    }
    __syncthreads();
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float gs = -a.gloss[0] / ((float)a.B * (float)P);
    const float* in = a.inner + (size_t)b * n * P + p;
    float* di = a.dinner + (size_t)b * n * P + p;
    const bool hard = a.alpha != a.alpha;
    if (hard) {
        float best = -INFINITY; int arg = 0;
        for (int c = 0; c < n; ++c) { const float ip = in[(size_t)c * P]; if (ip > best) { best = ip; arg = c; } }
        for (int c = 0; c < n; ++c) di[(size_t)c * P] = c == arg ? gs : 0.f;
    } else {
        float mx = -INFINITY;
        for (int c = 0; c < n; ++c) mx = fmaxf(mx, in[(size_t)c * P] * a.alpha);
        float z = 0.f, e1 = 0.f;
        for (int c = 0; c < n; ++c) { const float ip = in[(size_t)c * P], w = expf(ip * a.alpha - mx); z += w; e1 = fmaf(w, ip, e1); }
        const float mean_i = e1 / z;
        for (int c = 0; c < n; ++c) {
            const float ip = in[(size_t)c * P], pr = expf(ip * a.alpha - mx) / z;
            di[(size_t)c * P] = gs * pr * (1.f + a.alpha * (ip - mean_i));
        }
    }
    if (a.grad_x) {
        const float* x = a.x + (size_t)b * D * P + p;
        float s = 0.f;
        for (int d = 0; d < D; ++d) { const float v = x[(size_t)d * P]; s = fmaf(v, v, s); }
        const float nrm = fmaxf(sqrtf(s), DG_NORM_EPS), inv = 1.f / nrm;
        // d nf[d] = sum_c dinner[c] nc[c][d];   d x = (d nf - nf <nf, d nf>) / ||x||   (||x|| above eps)
        float dot = 0.f;
        for (int d = 0; d < D; ++d) {
            float g = 0.f;
            for (int c = 0; c < n; ++c) g = fmaf(di[(size_t)c * P], nc[c * ld + d], g);
            dot = fmaf(g, x[(size_t)d * P] * inv, dot);
        }
        const bool clampd = sqrtf(s) < DG_NORM_EPS;
        for (int d = 0; d < D; ++d) {
            float g = 0.f;
            for (int c = 0; c < n; ++c) g = fmaf(di[(size_t)c * P], nc[c * ld + d], g);
            a.grad_x[((size_t)b * D + d) * P + p] = clampd ? g * inv : (g - x[(size_t)d * P] * inv * dot) * inv;
        }
    }
}

// partial d loss / d nc: block = (chunk of 64 positions, image); dinner and the normalised x of the chunk in LDS, every thread a
// few (centre, channel) outputs
__global__ __launch_bounds__(256) void k_cluster_bwd_centres(const DgClusterBwdArgs a) {
    extern __shared__ float csm[];
    const int D = a.D, n = a.n, P = a.P;
    float* sdi = csm;                 // [n][65]
    float* snf = csm + n * 65;        // [D][65]
    __shared__ float sinv[64];
    const int b = blockIdx.y, p0 = blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const int p = p0 + threadIdx.x;
        float s = 0.f;
        if (p < P) for (int d = 0; d < D; ++d) { const float v = a.x[((size_t)b * D + d) * P + p]; s = fmaf(v, v, s); }
        sinv[threadIdx.x] = 1.f / fmaxf(sqrtf(s), DG_NORM_EPS);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * 64; i += 256) {
        const int c = i >> 6, q = i & 63;
        sdi[c * 65 + q] = p0 + q < P ? a.dinner[((size_t)b * n + c) * P + p0 + q] : 0.f;
    }
    for (int i = threadIdx.x; i < D * 64; i += 256) {
        const int d = i >> 6, q = i & 63;
        snf[d * 65 + q] = p0 + q < P ? a.x[((size_t)b * D + d) * P + p0 + q] * sinv[q] : 0.f;
    }
    __syncthreads();
    float* out = a.part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * n * D;
    for (int o = threadIdx.x; o < n * D; o += 256) {
        const int c = o / D, d = o - c * D;
        float s = 0.f;
        for (int q = 0; q < 64; ++q) s = fmaf(sdi[c * 65 + q], snf[d * 65 + q], s);
        out[o] = s;
    }
}

// d clusters[c] = (g - nc <nc, g>) / ||c||, g = sum of the partial sums (fixed order); one block per centre
__global__ __launch_bounds__(128) void k_cluster_bwd_finish(const DgClusterBwdArgs a, int nparts) {
    __shared__ float sg[128], sdot[128];
    const int c = blockIdx.x, d = threadIdx.x, D = a.D, n = a.n;
    float g = 0.f, v = 0.f;
    if (d < D) {
        for (int k = 0; k < nparts; ++k) g += a.part[((size_t)k * n + c) * D + d];
        v = a.clusters[(size_t)c * D + d];
    }
    sg[d] = v * v; sdot[d] = 0.f;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) { if (d < o) sg[d] += sg[d + o]; __syncthreads(); }
    const float raw = sqrtf(sg[0]), nrm = fmaxf(raw, DG_NORM_EPS);
    __syncthreads();
    sdot[d] = g * (v / nrm);
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) { if (d < o) sdot[d] += sdot[d + o]; __syncthreads(); }
    if (d < D) a.grad_clusters[(size_t)c * D + d] = raw < DG_NORM_EPS ? g / nrm : (g - (v / nrm) * sdot[0]) / nrm;
}

hipError_t dg_launch_cluster_bwd(const DgClusterBwdArgs& a, hipStream_t s) {
    const int smem1 = a.n * (a.D + 1) * 4, smem2 = (a.n + a.D) * 65 * 4;
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_cluster_bwd_pos), smem1);
    if (e != hipSuccess) return e;
    e = dg_set_max_smem(reinterpret_cast<const void*>(k_cluster_bwd_centres), smem2);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cluster_bwd_pos, dim3((a.P + 255) / 256, a.B), dim3(256), smem1, s, a);
    const int nbx = (a.P + 63) / 64;
    hipLaunchKernelGGL(k_cluster_bwd_centres, dim3(nbx, a.B), dim3(256), smem2, s, a);
    hipLaunchKernelGGL(k_cluster_bwd_finish, dim3(a.n), dim3(128), 0, s, a, nbx * a.B);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ linear probe: resize + CE

// source coordinate of F.interpolate(mode='bilinear', align_corners=False): max((dst + 0.5) * in / out - 0.5, 0)
__device__ __forceinline__ void resize_taps(const int dst, const int in, const int out, int& i0, int& i1, float& l1) {
    const float scale = (float)in / (float)out;
    float src = ((float)dst + 0.5f) * scale - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src; if (i0 > in - 1) i0 = in - 1;
    i1 = i0 < in - 1 ? i0 + 1 : i0;
    l1 = src - (float)i0;
}

// forward: block = one label row (b, Y); the two source rows of the logits sit in LDS
__global__ __launch_bounds__(256) void k_probe_ce_fwd(const DgProbeCeArgs a) {
    extern __shared__ float lsm[];                    // [2][n][w]
    __shared__ float red[8];
    const int b = blockIdx.y, Y = blockIdx.x, n = a.n, w = a.w, h = a.h;
    int y0, y1; float ly;
    resize_taps(Y, h, a.H, y0, y1, ly);
    for (int i = threadIdx.x; i < 2 * n * w; i += 256) {
        const int r = i / (n * w), c = (i / w) % n, x = i % w;
        lsm[i] = a.logits[(((size_t)b * n + c) * h + (r ? y1 : y0)) * w + x];
    }
    __syncthreads();
    float lsum = 0.f, cnt = 0.f;
    for (int X = threadIdx.x; X < a.W; X += 256) {
        const int64_t lab = a.label[((size_t)b * a.H + Y) * a.W + X];
        if (lab < 0 || lab >= n) continue;
        int x0, x1; float lx;
        resize_taps(X, w, a.W, x0, x1, lx);
        // a pixel's resized logits are re-blended per pass (six multiply-adds each) instead of being kept in an indexed array
        auto val = [&](const int c) {
            const float* r0 = lsm + c * w, *r1 = lsm + (n + c) * w;
            const float top = r0[x0] * (1.f - lx) + r0[x1] * lx, bot = r1[x0] * (1.f - lx) + r1[x1] * lx;
            return top * (1.f - ly) + bot * ly;
        };
        float mx = -INFINITY, z = 0.f;
        for (int c = 0; c < n; ++c) mx = fmaxf(mx, val(c));
        for (int c = 0; c < n; ++c) z += expf(val(c) - mx);
        const float vl = val((int)lab);
        lsum += logf(z) + mx - vl;
        cnt += 1.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lsum += __shfl_xor(lsum, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lsum; red[4 + (threadIdx.x >> 6)] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = a.part + ((size_t)b * a.H + Y) * 2;
        o[0] = red[0] + red[1] + red[2] + red[3];
        o[1] = red[4] + red[5] + red[6] + red[7];
    }
}

// out = {sum of part[i][0], sum of part[i][1], loss = sum / count}
__global__ __launch_bounds__(64) void k_probe_ce_finish(const float* __restrict__ part, int rows, float* __restrict__ out) {
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < rows; i += 64) { s += part[2 * i]; c += part[2 * i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if (threadIdx.x == 0) { out[0] = s; out[1] = c; out[2] = s / c; }
}

hipError_t dg_launch_probe_ce_fwd(const DgProbeCeArgs& a, float* out3, hipStream_t s) {
    const int smem = 2 * a.n * a.w * 4;
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_probe_ce_fwd), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_probe_ce_fwd, dim3(a.H, a.B), dim3(256), smem, s, a);
    hipLaunchKernelGGL(k_probe_ce_finish, dim3(1), dim3(64), 0, s, a.part, a.B * a.H, out3);
    return hipGetLastError();
}

// backward: block = one LOW-resolution row (b, y): walks the label rows whose resize touches y, per row the softmax minus one-hot of
// every labelled pixel goes to LDS ([n][W]), then each (class, low-res column) thread gathers the pixels that touch its column.
__global__ __launch_bounds__(256) void k_probe_ce_bwd(const DgProbeCeArgs a) {
    extern __shared__ float lsm[];
    const int b = blockIdx.y, y = blockIdx.x, n = a.n, w = a.w, h = a.h, H = a.H, W = a.W;
    float* rows = lsm;                    // [2][n][w] logits of the two source rows of the current label row
    float* dpx = lsm + 2 * n * w;         // [n][W] d loss / d resized logits of the current label row (times the row weight)
    const float gs = a.gloss[0] / a.total[1];
    const int nout = n * w;
    float accv[8];                        // outputs o = tid + 256 k  (n * w <= 2048)
#pragma unroll
    for (int k = 0; k < 8; ++k) accv[k] = 0.f;
    // label rows Y with y in {y0(Y), y1(Y)}: a contiguous range around (y + 0.5) * H / h
    const float up = (float)H / (float)h;
    int Ylo = (int)floorf(((float)y - 1.0f) * up) - 1, Yhi = (int)ceilf(((float)y + 2.0f) * up) + 1;
    Ylo = Ylo < 0 ? 0 : Ylo; Yhi = Yhi > H - 1 ? H - 1 : Yhi;
    for (int Y = Ylo; Y <= Yhi; ++Y) {
        int y0, y1; float ly;
        resize_taps(Y, h, H, y0, y1, ly);
        float wy = 0.f;
        if (y0 == y) wy += 1.f - ly;
        if (y1 == y) wy += ly;
        if (wy == 0.f) continue;                                   // (block-uniform)
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * n * w; i += 256) {
            const int r = i / (n * w), c = (i / w) % n, x = i % w;
            rows[i] = a.logits[(((size_t)b * n + c) * h + (r ? y1 : y0)) * w + x];
        }
        __syncthreads();
        for (int X = threadIdx.x; X < W; X += 256) {
            const int64_t lab = a.label[((size_t)b * H + Y) * W + X];
            const bool ok = lab >= 0 && lab < n;
            int x0, x1; float lx;
            resize_taps(X, w, W, x0, x1, lx);
            auto val = [&](const int c) {
                const float* r0 = rows + c * w, *r1 = rows + (n + c) * w;
                const float top = r0[x0] * (1.f - lx) + r0[x1] * lx, bot = r1[x0] * (1.f - lx) + r1[x1] * lx;
                return top * (1.f - ly) + bot * ly;
            };
            float mx = -INFINITY, z = 1.f;
            if (ok) {
                z = 0.f;
                for (int c = 0; c < n; ++c) mx = fmaxf(mx, val(c));
                for (int c = 0; c < n; ++c) z += expf(val(c) - mx);
            }
            const float sc = gs * wy / z;
            for (int c = 0; c < n; ++c)
                dpx[c * W + X] = ok ? (expf(val(c) - mx) - (c == (int)lab ? z : 0.f)) * sc : 0.f;
        }
        __syncthreads();
        // gather along the row: output (class c, column x) sums the pixels X whose taps include x
        const float upx = (float)W / (float)w;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int o = threadIdx.x + 256 * k;
            if (o >= nout) break;
            const int c = o / w, x = o - c * w;
            int Xlo = (int)floorf(((float)x - 1.0f) * upx) - 1, Xhi = (int)ceilf(((float)x + 2.0f) * upx) + 1;
            Xlo = Xlo < 0 ? 0 : Xlo; Xhi = Xhi > W - 1 ? W - 1 : Xhi;
            float s = 0.f;
            for (int X = Xlo; X <= Xhi; ++X) {
                int x0, x1; float lx;
                resize_taps(X, w, W, x0, x1, lx);
                float wx = 0.f;
                if (x0 == x) wx += 1.f - lx;
                if (x1 == x) wx += lx;
                s = fmaf(dpx[c * W + X], wx, s);
            }
            accv[k] += s;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int o = threadIdx.x + 256 * k;
        if (o >= nout) break;
        const int c = o / w, x = o - c * w;
        a.grad_logits[(((size_t)b * n + c) * h + y) * w + x] = accv[k];
    }
}

hipError_t dg_launch_probe_ce_bwd(const DgProbeCeArgs& a, hipStream_t s) {
    const int smem = (2 * a.n * a.w + a.n * a.W) * 4;
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_probe_ce_bwd), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_probe_ce_bwd, dim3(a.h, a.B), dim3(256), smem, s, a);
    return hipGetLastError();
}

// Coordinate samplers of the loss other than FPS, one block per image, no host sync:
//   k_salience_coords  sample_nonzero_locations            (reference src/modules.py:1191-1204)
//   k_simple_coords    simple_depth_informed_sampling      (reference src/modules.py:828-883)
// The reference draws integer ranks with torch.randint / torch.multinomial on the host side of a .nonzero() sync; here
// the caller hands over iid uniforms u in [0,1) (torch.rand on the device) and a rank among `count` candidates is
// min(int(u * count), count - 1) - the same distribution, and a deterministic function of (input, u) that the parity
// tests check pixel for pixel.
#include "dg_common.h"

#define SMP_THREADS 1024

__device__ __forceinline__ int rank_from_uniform(float u, int count) {
    const int r = (int)(u * (float)count);
    return min(max(r, 0), count - 1);
}

// salience (B,H,W): position s of image b gets the rank-th non-zero pixel in row-major order (torch.nonzero order),
// as (x, y) = (col / H, row / H) * 2 - 1: the reference divides BOTH coordinates by t.shape[1] and flips the pair
// (src/modules.py:1202-1204).  An image without non-zeros gets uniform integers in [0,H) for both (modules.py:1198).
__global__ __launch_bounds__(SMP_THREADS) void k_salience_coords(const float* __restrict__ sal, int H, int W, int n,
                                                                 const float* __restrict__ u_sel,
                                                                 const float* __restrict__ u_fb, float* __restrict__ out) {
    __shared__ int pre[SMP_THREADS + 1];        // exclusive prefix of the per-thread non-zero counts
    __shared__ int wtot[SMP_THREADS / 64];
    const int tid = threadIdx.x, b = blockIdx.x, HW = H * W;
    const float* img = sal + (size_t)b * HW;
    const int per = (HW + SMP_THREADS - 1) / SMP_THREADS;      // every thread owns a contiguous run of pixels
    const int b0 = min(tid * per, HW), b1 = min(b0 + per, HW);
    int cnt = 0;
    for (int i = b0; i < b1; ++i) cnt += img[i] != 0.f;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
    if ((tid & 63) == 63) wtot[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int wv = 0; wv < (tid >> 6); ++wv) base += wtot[wv];
    pre[tid] = base + incl - cnt;
    if (tid == SMP_THREADS - 1) pre[SMP_THREADS] = base + incl;
    __syncthreads();
    const int total = pre[SMP_THREADS];
    const float fH = (float)H;
    for (int s = tid; s < n; s += SMP_THREADS) {
        const size_t o = (size_t)b * n + s;
        int y, x;
        if (total == 0) {
            y = rank_from_uniform(u_fb[2 * o], H);
            x = rank_from_uniform(u_fb[2 * o + 1], H);
        } else {
            const int r = rank_from_uniform(u_sel[o], total);
            int lo = 0, hi = SMP_THREADS - 1;                 // last thread whose prefix is <= r owns rank r
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= r) lo = mid; else hi = mid - 1; }
            int k = r - pre[lo], i = min(lo * per, HW);
            for (;; ++i) if (img[i] != 0.f) { if (k == 0) break; --k; }
            y = i / W; x = i - y * W;
        }
        out[2 * o] = ((float)x / fH) * 2.f - 1.f;
        out[2 * o + 1] = ((float)y / fH) * 2.f - 1.f;
    }
}

// depth (B,1,H,W) -> adaptive_max_pool2d to (h,w) -> round to one decimal -> sort the pixels by (value, row-major index).
// Sample s: u_val picks a sorted position R (a value with probability count/HW, what the reference's multinomial over
// torch.unique counts does, modules.py:839-849), u_pick the rank inside the run of pixels sharing that value
// (modules.py:860-864, torch.nonzero order).  out (B,n,1,2) = ((row + .5) / h, (col + .5) / w) * 2 - 1  (modules.py:872, 1300).
__global__ __launch_bounds__(SMP_THREADS) void k_simple_coords(const float* __restrict__ depth, int H, int W, int h, int w,
                                                               int n, int NP2, const float* __restrict__ u_val,
                                                               const float* __restrict__ u_pick, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // [NP2] (sortable value bits << 32) | pixel
    const int tid = threadIdx.x, b = blockIdx.x, HW = h * w;
    const float* img = depth + (size_t)b * H * W;
    for (int p = tid; p < NP2; p += SMP_THREADS) {
        unsigned long long key = ~0ull;
        if (p < HW) {
            const int i = p / w, j = p - i * w;
            const int ys = (i * H) / h, ye = ((i + 1) * H + h - 1) / h;
            const int xs = (j * W) / w, xe = ((j + 1) * W + w - 1) / w;
            float m = -__builtin_inff();
            for (int y = ys; y < ye; ++y)
                for (int x = xs; x < xe; ++x) m = fmaxf(m, img[(size_t)y * W + x]);
            const float q = rintf(m * 10.f) / 10.f + 0.f;            // (depth * 10).round() / 10; -0 -> +0 (unique: equal)
            const unsigned int bits = __float_as_uint(q);
            const unsigned int ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
            key = ((unsigned long long)ord << 32) | (unsigned int)p;
        }
        keys[p] = key;
    }
    __syncthreads();
    for (int k = 2; k <= NP2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NP2; i += SMP_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], c = keys[ixj];
                    if ((a > c) == ((i & k) == 0)) { keys[i] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    const float fh = (float)h, fw = (float)w;
    for (int s = tid; s < n; s += SMP_THREADS) {
        const size_t o = (size_t)b * n + s;
        const int R = rank_from_uniform(u_val[o], HW);
        const unsigned int v = (unsigned int)(keys[R] >> 32);
        int lo = 0, hi = R;                                   // first position of the run of equal values
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((unsigned int)(keys[mid] >> 32) < v) lo = mid + 1; else hi = mid; }
        const int start = lo;
        lo = R; hi = HW;                                      // one past its last position
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((unsigned int)(keys[mid] >> 32) <= v) lo = mid + 1; else hi = mid; }
        const int t = rank_from_uniform(u_pick[o], lo - start);
        const int pix = (int)(keys[start + t] & 0xffffffffu);
        const int row = pix / w, col = pix - row * w;
        out[2 * o] = (((float)row + 0.5f) / fh) * 2.f - 1.f;
        out[2 * o + 1] = (((float)col + 0.5f) / fw) * 2.f - 1.f;
    }
}

hipError_t dg_launch_salience_coords(const float* sal, int B, int H, int W, int n, const float* u_sel, const float* u_fb,
                                     float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_salience_coords, dim3(B), dim3(SMP_THREADS), 0, s, sal, H, W, n, u_sel, u_fb, out);
    return hipGetLastError();
}

hipError_t dg_launch_simple_coords(const float* depth, int B, int H, int W, int h, int w, int n, const float* u_val,
                                   const float* u_pick, float* out, hipStream_t s) {
    int np2 = 1;
    while (np2 < h * w) np2 <<= 1;
    hipLaunchKernelGGL(k_simple_coords, dim3(B), dim3(SMP_THREADS), (size_t)np2 * 8, s, depth, H, W, h, w, n, np2, u_val,
                       u_pick, out);
    return hipGetLastError();
}

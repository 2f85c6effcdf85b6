// developer micro-benchmark (route (a)): do lines WRITTEN by one kernel stay in the 256-MB Infinity Cache for
// the next kernel to read?  Kernel A writes X MB (default or non-temporal stores), kernel B reads the same X MB back (default or
// non-temporal loads); B's bandwidth against (i) the same read after a 1-GiB sweep has flushed the caches ("cold") and (ii) a
// re-read of what B itself just read ("read-warm").  If write->read >= 7 TB/s at 128 MB, k_corr2 -> k_gs can be split by pair-set
// groups so that each G chunk is consumed from cache.
// build: hipcc -O3 --offload-arch=gfx950 -o wr_rd_cache wr_rd_cache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <bool NT> __global__ __launch_bounds__(256) void k_write(v4i* __restrict__ dst, size_t n, int seed) {
    const size_t stride = (size_t)gridDim.x * 256;
    const v4i v = {seed, seed + 1, (int)threadIdx.x, (int)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
template <bool NT> __global__ __launch_bounds__(256) void k_read(const v4i* __restrict__ src, size_t n, int* out) {
    v4i acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        v4i t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= t[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
int main() {
    const size_t big = (size_t)1 << 30;
    v4i *buf, *flush; int* o;
    hipMalloc(&buf, (size_t)512 << 20); hipMalloc(&flush, big); hipMalloc(&o, 4);
    hipMemset(flush, 1, big);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 2048;
    auto flush_caches = [&]() { hipLaunchKernelGGL((k_read<false>), dim3(blocks), dim3(256), 0, 0, flush, big / 16, o); };
    // warm the clocks
    for (int r = 0; r < 20; ++r) flush_caches();
    hipDeviceSynchronize();
    printf("%6s %8s %8s | %10s %10s %10s\n", "MB", "store", "load", "wr->rd", "cold rd", "rd->rd");
    for (int mb : {32, 64, 96, 128, 192, 256, 288, 384}) {
        const size_t n = ((size_t)mb << 20) / 16;
        for (int wnt = 0; wnt < 2; ++wnt) for (int rnt = 0; rnt < 2; ++rnt) {
            auto wr = [&](int s) { if (wnt) hipLaunchKernelGGL((k_write<true>), dim3(blocks), dim3(256), 0, 0, buf, n, s);
                                   else hipLaunchKernelGGL((k_write<false>), dim3(blocks), dim3(256), 0, 0, buf, n, s); };
            auto rd = [&]() { if (rnt) hipLaunchKernelGGL((k_read<true>), dim3(blocks), dim3(256), 0, 0, buf, n, o);
                              else hipLaunchKernelGGL((k_read<false>), dim3(blocks), dim3(256), 0, 0, buf, n, o); };
            float t_wr = 0, t_cold = 0, t_rr = 0, ms;
            const int reps = 5;
            for (int r = 0; r < reps; ++r) {
                flush_caches(); wr(r);
                hipEventRecord(a); rd(); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); t_wr += ms;
                hipEventRecord(a); rd(); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); t_rr += ms;
                flush_caches();
                hipEventRecord(a); rd(); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); t_cold += ms;
            }
            auto tb = [&](float t) { return (double)mb * 1048576.0 * reps / (t * 1e-3) / 1e12; };
            printf("%6d %8s %8s | %7.2f TB/s %7.2f TB/s %7.2f TB/s\n", mb, wnt ? "nt" : "default", rnt ? "nt" : "default", tb(t_wr), tb(t_cold), tb(t_rr));
        }
    }
    return 0;
}

// micro-benchmark: sustained bf16 MFMA throughput of the two shapes k_corr2 could be tiled for, at the clock
// the chip HOLDS under load (MI355X_MICROARCH.md "DVFS give-back" item 7), with the kernel's operand pattern: the streamed
// operand's fragments come from LDS by ds_read_b128 (one fragment feeds two row fragments' worth of MFMAs), the stationary
// operand's fragments stay in registers, one wave per SIMD, every CU busy, random operands.
//   mode 0: v_mfma_f32_32x32x16_bf16, 64 rows x 32 streamed positions per wave: per k-step of 16 one LDS fragment, 2 MFMAs
//   mode 1: v_mfma_f32_16x16x32_bf16, same output tile: per k-step of 32 two LDS fragments, 8 MFMAs (4 row x 2 streamed sub-blocks)
// Both issue the same flops per LDS byte and per register.  Reports wall TFLOP/s over >= 1.5 s of back-to-back launches and the
// in-kernel clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shape_sustained mfma_shape_sustained.hip && ./mfma_shape_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <utility>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int NKS = 24;            // k-steps of 16 per chain (C = 384)
constexpr int TILE_BYTES = NKS * 1024;

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }
template <int I> __device__ __forceinline__ void agpr_put(const v4i& v) {       // a[4I..4I+3] <- v
    asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "n"(4 * I), "n"(4 * I + 1), "n"(4 * I + 2), "n"(4 * I + 3));
}
template <int OFF> __device__ __forceinline__ void lds_rd(v4i& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void wait_lgkm(void) { asm volatile("s_waitcnt lgkmcnt(%c0)" :: "n"(N) : "memory"); }
template <int I, bool ZERO> __device__ __forceinline__ void mfma32(f32x16& acc, const v4i& a) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(a), "n"(4 * I), "n"(4 * I + 3));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "n"(4 * I), "n"(4 * I + 3));
}
template <int I, bool ZERO> __device__ __forceinline__ void mfma16(f32x4& acc, const v4i& a) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(a), "n"(4 * I), "n"(4 * I + 3));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "n"(4 * I), "n"(4 * I + 3));
}
#define DG_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"

// The inner loop is hand-placed like k_corr2's: LDS reads run PF fragments ahead in a register ring behind counted lgkmcnt waits,
// the stationary fragments are literal accumulator-file registers a[0:191], accumulators are architectural VGPRs and restart
// from 0 at every tile (the first MFMA of a chain takes C = 0).
template <int MODE> __global__ __launch_bounds__(256) void k(const v4i* __restrict__ src, float* __restrict__ sink, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", DG_A8(1), DG_A8(2), DG_A8(3), DG_A8(4), DG_A8(5), DG_A8(6), DG_A8(7),
                 DG_A8(8), DG_A8(9), DG_A8(10), DG_A8(11), DG_A8(12), DG_A8(13), DG_A8(14), DG_A8(15), DG_A8(16), DG_A8(17), DG_A8(18), "a190", "a191");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * TILE_BYTES / 16; i += 256) reinterpret_cast<v4i*>(smem)[i] = src[(i * 7 + 3) & 4095];
    sfor<2 * NKS>([&](auto I) { agpr_put<I.value>(src[(wid * 64 + lane + 61 * I.value) & 4095]); });
    __syncthreads();
    constexpr int PF = 8;
    v4i ra[PF];
    const unsigned base = (unsigned)(size_t)smem + lane * 16;      // LDS byte address (the low 32 bits of a shared pointer)
    unsigned long long t0 = 0, r0 = 0;
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    float chk = 0.f;
    if constexpr (MODE == 0) {
        f32x16 acc0, acc1;
        for (int it = 0; it < iters; ++it) {
            const unsigned tile = base + (it & 1) * TILE_BYTES;
            sfor<PF>([&](auto I) { lds_rd<I.value * 1024>(ra[I.value], tile); });
            sfor<NKS>([&](auto S) {
                constexpr int s = S.value;
                constexpr int issued = s + PF < NKS ? s + PF : NKS;
                wait_lgkm<issued - (s + 1)>();
                mfma32<s, s == 0>(acc0, ra[s % PF]);
                mfma32<NKS + s, s == 0>(acc1, ra[s % PF]);
                if constexpr (s + PF < NKS) lds_rd<(s + PF) * 1024>(ra[s % PF], tile);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc0), "+v"(acc1));
        chk = acc0[0] + acc1[5];
    } else {
        f32x4 acc[8];
        for (int it = 0; it < iters; ++it) {
            const unsigned tile = base + (it & 1) * TILE_BYTES;
            sfor<PF>([&](auto I) { lds_rd<I.value * 1024>(ra[I.value], tile); });
            sfor<NKS / 2>([&](auto S) {
                constexpr int s = S.value;                      // k-step of 32: LDS fragments 2s (streamed sub-block 0) and 2s+1 (sub-block 1)
                constexpr int issued = 2 * s + PF < NKS ? 2 * s + PF : NKS;
                wait_lgkm<issued - (2 * s + 2)>();
                // four row sub-blocks (stationary registers: 16 bytes per sub-block and k-step of 32) x two streamed sub-blocks
                mfma16<4 * s + 0, s == 0>(acc[0], ra[(2 * s) % PF]);
                mfma16<4 * s + 0, s == 0>(acc[4], ra[(2 * s + 1) % PF]);
                mfma16<4 * s + 1, s == 0>(acc[1], ra[(2 * s) % PF]);
                mfma16<4 * s + 1, s == 0>(acc[5], ra[(2 * s + 1) % PF]);
                mfma16<4 * s + 2, s == 0>(acc[2], ra[(2 * s) % PF]);
                mfma16<4 * s + 2, s == 0>(acc[6], ra[(2 * s + 1) % PF]);
                mfma16<4 * s + 3, s == 0>(acc[3], ra[(2 * s) % PF]);
                mfma16<4 * s + 3, s == 0>(acc[7], ra[(2 * s + 1) % PF]);
                if constexpr (2 * s + PF < NKS) { lds_rd<(2 * s + PF) * 1024>(ra[(2 * s) % PF], tile); lds_rd<(2 * s + 1 + PF) * 1024>(ra[(2 * s + 1) % PF], tile); }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
        chk = acc[0][0] + acc[7][3] + acc[3][1] + acc[5][2];
    }
    if (chk == 1.2345f) sink[0] = chk;
    if (lane == 0 && wid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;      // tiles per wave and launch (~25 ms per launch)
    const double secs = argc > 2 ? atof(argv[2]) : 1.5;
    v4i* s; float* sink; unsigned long long* clk;
    hipMalloc(&s, 4096 * 16); hipMalloc(&sink, 64); hipMalloc(&clk, 256 * 16);
    {
        std::vector<unsigned short> hb(4096 * 8);
        unsigned x = 12345;
        for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 9) & 0x807f) | 0x3e80 | (((x >> 20) & 7) << 4)); }   // bf16 of magnitude 0.25 .. 2, random sign / mantissa
        if (getenv("ZERO")) std::fill(hb.begin(), hb.end(), 0);
        hipMemcpy(s, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES);
    const double flop_per_launch = 256.0 * 4 * iters * NKS * 2 * (2.0 * 32 * 32 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round)
        for (int mode = 0; mode < 2; ++mode) {
            // hold the load for `secs`, time the second half
            float ms = 0.f; int launches = 0; double total_ms = 0.0; std::vector<float> per;
            while (total_ms < secs * 1e3) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 2 * TILE_BYTES, 0, s, sink, clk, iters);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 2 * TILE_BYTES, 0, s, sink, clk, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                total_ms += ms; ++launches; per.push_back(ms);
            }
            std::vector<float> tail(per.begin() + per.size() / 2, per.end());
            std::sort(tail.begin(), tail.end());
            const float med = tail[tail.size() / 2];
            unsigned long long h[512]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
            std::vector<double> ghz, cyc;
            for (int b = 0; b < 256; ++b) { ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); cyc.push_back((double)h[2 * b]); }
            std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
            const double mfmas = (double)iters * NKS * 2 * (mode ? 2 : 1);
            printf("round %d %-22s %7.2f ms/launch  %7.1f TFLOP/s  in-kernel clock %.3f GHz  %.2f cycles per MFMA (x%d per 32x32x16-equivalent)\n", round,
                   mode ? "16x16x32 (8 acc)" : "32x32x16 (2 acc)", med, flop_per_launch / med * 1e-9, ghz[128], cyc[128] / mfmas, mode ? 2 : 1);
        }
    return 0;
}

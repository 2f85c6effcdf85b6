// developer micro-benchmark: how much do LDS-DMA writes (global_load_lds_dwordx4) landing in LDS slow down ds_read_b128 + MFMA
// chains of other waves on the same CU?  512-thread blocks; waves 0-3: 36 ds_read_b128 feeding 30 MFMAs per iteration;
// waves 4-7: mode 0 idle, mode 1: LDS-DMA of 36 KiB per iteration (9 pieces per wave), mode 2: global_load -> ds_write_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(512) void k(float* out, const char* src, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][36 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 36864 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 1023);
    __syncthreads();
    bf16x8 b[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.01f * ((tid + i + j) & 15));
    f32x16 acc = {};
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)smem);
    const char* gs = src + (size_t)blockIdx.x * 36864 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (wid < 4) {
            const char* base = smem + (it & 1) * 36864 + lane * 16;
            v4i ra[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) ra[i] = *reinterpret_cast<const v4i*>(base + i * 1024);
#pragma unroll
            for (int k2 = 0; k2 < 30; ++k2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[k2 % 6]), b[k2 & 7], acc, 0, 0, 0);
                if (k2 + 6 < 36) ra[k2 % 6] = *reinterpret_cast<const v4i*>(base + (k2 + 6) * 1024);
            }
        } else if (mode == 1) {
            const uint32_t dst = lds0 + ((it + 1) & 1) * 36864;
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) { const int c = (wid - 4) + 4 * kk; dma16(gs + c * 1024, dst + c * 1024); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (mode == 2) {
            char* dst = smem + ((it + 1) & 1) * 36864 + lane * 16;
            v4i t[9];
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) { const int c = (wid - 4) + 4 * kk; t[kk] = *reinterpret_cast<const v4i*>(gs + c * 1024); }
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) { const int c = (wid - 4) + 4 * kk; *reinterpret_cast<v4i*>(dst + c * 1024) = t[kk]; }
        }
        __builtin_amdgcn_s_barrier();
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * 512 + tid] = r;
}
int main() {
    float* out; char* src;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&src, 1024 * 36864); hipMemset(src, 0, 1024 * 36864);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 36864);
    const int iters = 2000, blocks = 1024;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 2 * 36864, 0, out, src, 10, mode);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 2 * 36864, 0, out, src, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %8.3f ms -> %6.0f cycles per iteration (2.1 GHz, 2 blocks/CU co-resident)\n", mode,
               mode == 0 ? "reads+MFMA only" : mode == 1 ? "+ LDS-DMA 36 KiB/iter" : "+ global_load/ds_write 36 KiB/iter", ms, ms * 1e-3 * 2.1e9 / iters / 2);
    }
    return 0;
}

// developer micro-benchmark: MFMA issue rate of 8-wave blocks (2 waves per SIMD), operands in registers vs from LDS
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: registers only, 1: A operand from LDS through a 4-deep ring, 2: as 1 + 16 VALU per 6 MFMAs
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 b[24];
    for (int i = 0; i < 24; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.01f * (tid + i + j));
    for (int i = tid; i < 36864 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc = {};
    float s = 0.f;
    const char* base = smem + (lane & 31) * 768;
    int swz[8];
    for (int j = 0; j < 8; ++j) swz[j] = ((2 * j + (lane >> 5)) ^ (lane & 15)) << 4;   // same XOR swizzle as the product kernel
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 24; ++k2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(k2 + 1) % 24], b[k2], acc, 0, 0, 0);
        } else {
            v4i ra[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const v4i*>(base + swz[i & 7] + (i >> 3) * 256);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k2 = 0; k2 < 24; ++k2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[k2 % 4]), b[k2], acc, 0, 0, 0);
                if (k2 + 4 < 24) ra[k2 % 4] = *reinterpret_cast<const v4i*>(base + swz[(k2 + 4) & 7] + ((k2 + 4) >> 3) * 256);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { float v = acc[i] + 0.5f; s += v >= 0.f ? v * acc[i] : 0.f; }
            }
        }
    }
    float r = s;
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * 512 + tid] = r;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 2000, blocks = 256 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 120 * 1024, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 120 * 1024, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 8 * iters * 24 * 32768.0;
    printf("%-28s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 512 * 4);
    run<0>("registers only", out);
    run<1>("A from LDS, ring of 4", out);
    run<2>("A from LDS + 16-elem VALU", out);
    return 0;
}

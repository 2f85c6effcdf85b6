// developer micro-benchmark: do an MFMA-chain wave and a VALU wave on the SAME SIMD overlap?
// 512-thread blocks (2 waves per SIMD).  role bits: 1 = waves 0-3 run the MFMA chain, 2 = waves 4-7 run the VALU block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters, int roles, int valu_first_half) {
    const int tid = threadIdx.x, wid = tid >> 6;
    const bool first = wid < 4;
    bf16x8 b[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.01f * ((tid + i + j) & 15));
    f32x16 acc = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (tid + i);
    const bool do_mfma = first ? (roles & 1) : (roles & 4);
    const bool do_valu = first ? (roles & 8) : (roles & 2);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int k2 = 0; k2 < 30; ++k2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[k2 & 7], b[(k2 + 1) & 7], acc, 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < NV / 16; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        }
        __builtin_amdgcn_s_barrier();
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i] + v[i];
    out[blockIdx.x * 512 + tid] = r;
}

int main() {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    const int iters = 2000, blocks = 1024;
    auto run = [&](const char* name, int roles) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<160>, dim3(blocks), dim3(512), 0, 0, out, 10, roles, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<160>, dim3(blocks), dim3(512), 0, 0, out, iters, roles, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 4 rounds of blocks per CU
        printf("%-44s %8.3f ms  -> %6.0f cycles per iteration per block (at 2.1 GHz)\n", name, ms, ms * 1e-3 * 2.1e9 / iters / 4);
    };
    run("waves0-3 MFMA(30) only", 1);
    run("waves4-7 VALU(160) only", 2);
    run("waves0-3 MFMA + waves4-7 VALU", 3);
    run("all 8 waves MFMA", 5);
    run("all 8 waves VALU", 10);
    run("all 8 waves MFMA then VALU (lockstep)", 15);
    return 0;
}

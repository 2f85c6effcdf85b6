// developer micro-benchmark: the shader clock a LIGHTLY loaded chip holds (few workgroups, latency-bound code), and the cost of
// the building blocks of a barrier-per-round loop.  build: hipcc -O3 --offload-arch=gfx950 -o clock_light_load clock_light_load.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int mode>
__global__ __launch_bounds__(256) void k(int rounds, unsigned long long* out, float* sink) {
    __shared__ float sl[2][8];
    float v = threadIdx.x * 1e-3f;
    float u[8] = {v, v, v, v, v, v, v, v};
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w[8];
    for (int j = 0; j < 8; ++j) w[j] = f2{v, v};
    if (threadIdx.x < 16) sl[threadIdx.x >> 3][threadIdx.x & 7] = 0.f;
    __syncthreads();
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < rounds; ++i) {
        if constexpr (mode == 0) {                 // 64 dependent VALU operations
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
        } else if constexpr (mode == 1) {          // LDS write, barrier, LDS read
            if ((threadIdx.x & 63) == 0) sl[i & 1][threadIdx.x >> 6] = v;
            __syncthreads();
            v += sl[i & 1][0] + sl[i & 1][1] + sl[i & 1][2] + sl[i & 1][3];
        } else if constexpr (mode == 2) {          // six dependent DPP steps + readlane
#pragma unroll
            for (int j = 0; j < 6; ++j) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false)));
            v += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
        } else if constexpr (mode == 3) {          // barrier only
            __syncthreads();
        } else if constexpr (mode == 4) {          // 64 VALU operations in 8 independent chains
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\t"
                             "v_add_f32 %4, %4, %4\n\tv_add_f32 %5, %5, %5\n\tv_add_f32 %6, %6, %6\n\tv_add_f32 %7, %7, %7"
                             : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
        } else if constexpr (mode == 5) {          // 64 packed fp32 multiplies in 8 independent chains
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_pk_mul_f32 %0, %0, %0\n\tv_pk_mul_f32 %1, %1, %1\n\tv_pk_mul_f32 %2, %2, %2\n\tv_pk_mul_f32 %3, %3, %3\n\t"
                             "v_pk_mul_f32 %4, %4, %4\n\tv_pk_mul_f32 %5, %5, %5\n\tv_pk_mul_f32 %6, %6, %6\n\tv_pk_mul_f32 %7, %7, %7"
                             : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
        } else if constexpr (mode == 6) {          // six single-instruction DPP max steps + readlane (as k_fps_coords)
            int iv = __float_as_int(v);
#pragma unroll
            for (int j = 0; j < 6; ++j) iv = max(iv, __builtin_amdgcn_update_dpp((int)0x80000000, iv, 0x111, 0xf, 0xf, false));
            v = __int_as_float(iv + __builtin_amdgcn_readlane(iv, 63));
        } else if constexpr (mode == 7) { // LDS broadcast read that depends on the previous one (address from the loaded value)
            v = sl[0][__float_as_int(v) & 7];
        } else if constexpr (mode == 8) { // empty loop body
            asm volatile("" : "+v"(v));
        } else {                         // a uniform branch that is taken every other round around one instruction
            if (i & 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
            else asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v));
        }
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = w1 - w0; out[blockIdx.x * 2 + 1] = c1 - c0; }
    for (int j = 0; j < 8; ++j) v += u[j] + w[j][0] + w[j][1];
    if (v == 123.456f) sink[0] = v;
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 16, threads = argc > 2 ? atoi(argv[2]) : 256, rounds = 2000;
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 16); hipMalloc(&sink, 4);
    const char* names[] = {"64 dependent v_add_f32", "LDS write + barrier + 4 LDS reads", "6 DPP max steps + readlane", "barrier",
                           "64 v_add_f32, 8 chains", "64 v_pk_mul_f32, 8 chains", "6 one-instruction DPP steps + readlane", "dependent LDS read", "empty loop", "alternating uniform branch"};
    for (int mode = 0; mode < 10; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
#define L(M) case M: hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(threads), 0, 0, rounds, out, sink); break;
            switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) }
#undef L
        }
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        const double ns = h[0] * 10.0;      // wall_clock64: 100 MHz
        printf("%d blocks x %d threads  %-40s %8.1f ns/round  %8.1f shader cycles/round  -> %.2f GHz\n", blocks, threads, names[mode], ns / rounds, (double)h[1] / rounds, h[1] / ns);
    }
    return 0;
}

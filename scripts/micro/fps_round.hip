// developer micro-benchmark: the round loop of k_fps_coords (dg_post.hip) on synthetic points, with parts removable at build
// time, to see which part of the one-wave-per-SIMD dependent chain costs what.
//   hipcc -O3 --offload-arch=gfx950 [-DNO_POINTS] [-DNO_DPP] [-DNO_PUBLISH] [-DNO_BARRIER] [-DNO_COMBINE] [-DNO_RETIRE] -o fps_round fps_round.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define FPS_THREADS 256
typedef float fps2 __attribute__((ext_vector_type(2)));
template <bool HI> __device__ __forceinline__ fps2 fps_sub_bcast(fps2 l, fps2 q) {
    fps2 r;
    if constexpr (HI) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(l), "v"(q));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(l), "v"(q));
    return r;
}
__device__ __forceinline__ fps2 fps_mul(fps2 a, fps2 b) { fps2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ fps2 fps_add(fps2 a, fps2 b) { fps2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fps_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int dpp_wave_max_i(int v) {
    const int small = (int)0x80000000;
#define DG_DPP_MAXI(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(small, v, ctrl, rmask, 0xf, false))
    DG_DPP_MAXI(0x111, 0xf); DG_DPP_MAXI(0x112, 0xf); DG_DPP_MAXI(0x114, 0xf); DG_DPP_MAXI(0x118, 0xf);
    DG_DPP_MAXI(0x142, 0xa); DG_DPP_MAXI(0x143, 0xc);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int dpp_wave_min(int v) {
    const int big = 0x7fffffff;
#define DG_DPP_MIN(ctrl, rmask) v = min(v, __builtin_amdgcn_update_dpp(big, v, ctrl, rmask, 0xf, false))
    DG_DPP_MIN(0x111, 0xf); DG_DPP_MIN(0x112, 0xf); DG_DPP_MIN(0x114, 0xf); DG_DPP_MIN(0x118, 0xf);
    DG_DPP_MIN(0x142, 0xa); DG_DPP_MIN(0x143, 0xc);
    return __builtin_amdgcn_readlane(v, 63);
}
template <int NPT>
__global__ __launch_bounds__(FPS_THREADS) void k(int nsel, const float* __restrict__ pts, unsigned long long* out, int* order_out) {
    constexpr int NWV = FPS_THREADS / 64, NP2 = NPT / 2;
    __shared__ int order[1024];
    __shared__ __attribute__((aligned(16))) long long skey[2][NWV];
    __shared__ __attribute__((aligned(16))) float4 sxyz[2][NWV];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    fps2 qx[NP2], qy[NP2], qz[NP2], qd[NP2];
    for (int k = 0; k < NPT; ++k) {
        const int idx = tid + FPS_THREADS * k;
        qx[k >> 1][k & 1] = pts[idx * 3]; qy[k >> 1][k & 1] = pts[idx * 3 + 1]; qz[k >> 1][k & 1] = pts[idx * 3 + 2];
        qd[k >> 1][k & 1] = __builtin_inff();
    }
    if (tid == 0) { qd[0][0] = -1.f; sxyz[0][0] = make_float4(qx[0][0], qy[0][0], qz[0][0], 0.f); }
    if (tid < 8) (&skey[0][0])[tid] = tid;
    if (lane == 0) order[0] = 0;
    __syncthreads();
    fps2 lxy = {sxyz[0][0].x, sxyz[0][0].y}, lzz = {sxyz[0][0].z, 0.f};
    const unsigned long long w0 = wall_clock64();
    for (int it = 1; it < nsel; ++it) {
        float nd[NPT];
#ifndef NO_POINTS
#pragma unroll
        for (int p = 0; p < NP2; ++p) {
            const fps2 dx = fps_sub_bcast<false>(lxy, qx[p]), dy = fps_sub_bcast<true>(lxy, qy[p]), dz = fps_sub_bcast<false>(lzz, qz[p]);
            const fps2 dd = fps_add(fps_add(fps_mul(dx, dx), fps_mul(dy, dy)), fps_mul(dz, dz));
            qd[p][0] = nd[2 * p] = fps_min(dd[0], qd[p][0]);
            qd[p][1] = nd[2 * p + 1] = fps_min(dd[1], qd[p][1]);
        }
#else
        for (int k = 0; k < NPT; ++k) { nd[k] = qd[k >> 1][k & 1] + lxy[0]; asm volatile("" : "+v"(nd[k])); }
#endif
        int mb = __float_as_int(nd[0]);
#pragma unroll
        for (int k = 1; k < NPT; ++k) mb = max(mb, __float_as_int(nd[k]));
#ifndef NO_DPP
        const int wmax = dpp_wave_max_i(mb);
        unsigned long long m = __ballot(mb == wmax);
        if (__builtin_expect(__popcll(m) > 1, 0)) {
            int first = 0x7fffffff;
#pragma unroll
            for (int k = NPT - 1; k >= 0; --k) first = __float_as_int(nd[k]) == wmax ? tid + FPS_THREADS * k : first;
            const int imin = dpp_wave_min(first);
            m = __ballot(first == imin);
        }
        const int src = (int)__ffsll((long long)m) - 1;
#else
        const int src = it & 63;
#endif
        const int par = it & 1;
#ifndef NO_PUBLISH
        if (lane == src) {
            int bk = NPT - 1;
            float bx = qx[NP2 - 1][1], by = qy[NP2 - 1][1], bz = qz[NP2 - 1][1];
#pragma unroll
            for (int k = NPT - 2; k >= 0; --k) {
                const bool hit = __float_as_int(nd[k]) == mb;
                bk = hit ? k : bk; bx = hit ? qx[k >> 1][k & 1] : bx; by = hit ? qy[k >> 1][k & 1] : by; bz = hit ? qz[k >> 1][k & 1] : bz;
            }
            skey[par][wv] = ((long long)mb << 32) | (long long)(0x7fffffff - (tid + FPS_THREADS * bk));
            sxyz[par][wv] = make_float4(bx, by, bz, 0.f);
        }
#endif
#ifndef NO_BARRIER
        __syncthreads();
#endif
#ifndef NO_COMBINE
        long long k0 = skey[par][0], k1 = skey[par][1], k2 = skey[par][2], k3 = skey[par][3];
        const bool s01 = k1 > k0, s23 = k3 > k2;
        const long long k01 = s01 ? k1 : k0, k23 = s23 ? k3 : k2;
        const bool shi = k23 > k01;
        const long long kb = shi ? k23 : k01;
        const int wsel = shi ? (s23 ? 3 : 2) : (s01 ? 1 : 0);
        const float4 bp = sxyz[par][wsel];
        const int i0 = 0x7fffffff - (int)(kb & 0xffffffffll);
        lxy[0] = bp.x; lxy[1] = bp.y; lzz[0] = bp.z;
#else
        const int i0 = (mb >> 3) & 1023;
        lxy[0] += 1.f; lxy[1] += 2.f; lzz[0] += 3.f;
#endif
#ifndef NO_RETIRE
        const int rel = i0 - tid;
#pragma unroll
        for (int k = 0; k < NPT; ++k) qd[k >> 1][k & 1] = rel == FPS_THREADS * k ? -1.f : qd[k >> 1][k & 1];
#endif
        if (lane == 0) order[it] = i0;
    }
    const unsigned long long w1 = wall_clock64();
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = w1 - w0;
    float acc = 0.f;
    for (int p = 0; p < NP2; ++p) acc += qd[p][0] + qd[p][1];
    if (blockIdx.x == 0) { for (int k = tid; k < nsel; k += FPS_THREADS) order_out[k] = order[k]; if (acc == 12345.f) order_out[0] = 1; }
}
int main(int argc, char** argv) {
    const int blocks = 16, nsel = argc > 1 ? atoi(argv[1]) : 121;
    float* pts; unsigned long long* out; int* ord;
    hipMalloc(&pts, 1024 * 3 * 4); hipMalloc(&out, blocks * 8); hipMalloc(&ord, 4096);
    float h[1024 * 3];
    srand(1);
    for (int i = 0; i < 1024 * 3; ++i) h[i] = (rand() % 100000) * 1e-4f;
    hipMemcpy(pts, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(FPS_THREADS), 0, 0, nsel, pts, out, ord);
    hipDeviceSynchronize();
    unsigned long long t; int o[4];
    hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); hipMemcpy(o, ord, 16, hipMemcpyDeviceToHost);
    printf("%-60s %7.1f ns per round  (order %d %d %d %d)\n", argc > 2 ? argv[2] : "full", t * 10.0 / (nsel - 1), o[0], o[1], o[2], o[3]);
    return 0;
}

// developer check: v_dot2c_f32_bf16 against two fp32 fmas on random bf16 pairs (is it usable for squared norms?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const int* in, float* o1, float* o2, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f, b = 0.f;
    for (int r = 0; r < 8; ++r) {
        int v = in[i * 8 + r];
        bf2 p = __builtin_bit_cast(bf2, v);
        a = __builtin_amdgcn_fdot2_f32_bf16(p, p, a, false);
        float x0 = __builtin_bit_cast(float, v << 16), x1 = __builtin_bit_cast(float, v & (int)0xffff0000);
        b = fmaf(x0, x0, b); b = fmaf(x1, x1, b);
    }
    o1[i] = a; o2[i] = b;
}
int main() {
    const int n = 4096;
    int* h = (int*)malloc(n * 8 * 4);
    srand(1);
    for (int i = 0; i < n * 8; ++i) {
        float f0 = (rand() / (float)RAND_MAX - 0.5f) * ((i & 64) ? 4.f : 1e-3f), f1 = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
        unsigned u0, u1; memcpy(&u0, &f0, 4); memcpy(&u1, &f1, 4);
        h[i] = (int)((u0 >> 16) | (u1 & 0xffff0000u));
    }
    int* d; float *o1, *o2;
    hipMalloc(&d, n * 32); hipMalloc(&o1, n * 4); hipMalloc(&o2, n * 4);
    hipMemcpy(d, h, n * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o1, o2, n);
    float* a = (float*)malloc(n * 4); float* b = (float*)malloc(n * 4);
    hipMemcpy(a, o1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b, o2, n * 4, hipMemcpyDeviceToHost);
    double worst = 0; int wi = 0;
    for (int i = 0; i < n; ++i) { double e = fabs(a[i] - b[i]) / fmax(fabs(b[i]), 1e-30); if (e > worst) { worst = e; wi = i; } }
    printf("worst relative difference dot2 vs fma: %.3g (dot2 %.9g fma %.9g)\n", worst, a[wi], b[wi]);
    return 0;
}

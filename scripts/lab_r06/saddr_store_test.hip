// developer aid: does global_store_dwordx4 in the saddr form (scalar base + 32-bit lane offset, inst offset, nt) work from inline asm?
//   hipcc --offload-arch=gfx950 -O2 scripts/lab_r06/saddr_store_test.hip -o /tmp/saddr_test && /tmp/saddr_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int variant) {
    const int lane = threadIdx.x & 63;
    v4i d = {lane, lane + 100, lane + 200, lane + 300};
    uint64_t base = reinterpret_cast<uint64_t>(out) + (uint64_t)blockIdx.x * 4096;
    // (readfirstlane returns int: without the casts to uint32_t the low half is SIGN-extended into the high half - the fault of experiments/r06.md 18.12)
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((uint64_t)bhi << 32) | blo;
    const uint32_t off = lane * 16;
    if (variant == 0) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off), "v"(d), "s"(base) : "memory");
    if (variant == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(d), "s"(base) : "memory");
    if (variant == 2) asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024" :: "v"(off), "v"(d), "s"(base) : "memory");
    if (variant == 3) asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024 nt" :: "v"(off), "v"(d), "s"(base) : "memory");
}
int main() {
    int* buf;
    hipMalloc(&buf, 1 << 20);
    for (int v = 0; v < 4; ++v) {
        hipMemset(buf, 0xff, 1 << 20);
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, buf, v);
        hipError_t e = hipDeviceSynchronize();
        int h[2048];
        hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
        const int o = (v >= 2) ? 256 : 0;
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) if (h[o + 4 * l + c] != l + 100 * c) ++bad;
        printf("variant %d: %s, %d wrong of 256\n", v, hipGetErrorString(e), bad);
    }
    return 0;
}

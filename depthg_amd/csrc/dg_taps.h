// The bilinear tap map of sample() (src/modules.py:822-825: grid_sample, border, align_corners=True) and its inverse records - shared by
// dg_post.hip (the sample() adjoint, k_build_taps / k_pre_general) and dg_prep.hip (the same records as extra blocks of the gather launch).
#pragma once
#include "dg_common.h"

#ifdef __HIPCC__
__device__ __forceinline__ void dg_taps(const float* c, int h, int w, int& x0, int& y0, bool& inx, bool& iny,
                                        float& w00, float& w01, float& w10, float& w11) {
    float x = ((c[0] + 1.f) / 2.f) * (float)(w - 1);
    float y = ((c[1] + 1.f) / 2.f) * (float)(h - 1);
    x = fminf(fmaxf(x, 0.f), (float)(w - 1));
    y = fminf(fmaxf(y, 0.f), (float)(h - 1));
    const float x0f = floorf(x), y0f = floorf(y);
    const float wx1 = x - x0f, wy1 = y - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    x0 = (int)x0f; y0 = (int)y0f;
    inx = x0 + 1 <= w - 1; iny = y0 + 1 <= h - 1;
    w00 = wy0 * wx0; w01 = wy0 * wx1; w10 = wy1 * wx0; w11 = wy1 * wx1;
}

// Inverse of the bilinear tap map of sample(), once per (coords set, image): for every pixel the list of
// (position, weight) that read it, as CSR in global memory (off[HW+1], then 4P weights, then 4P positions).
// Lists are sorted by position, so the gather below sums in a fixed order (bit-reproducible gradients).
// grid (B, 2), block SCAT_THREADS, dynamic LDS (2*HW + 1) ints + 4P floats + 4P ushorts.
struct DgTapsArgs { const float* coords1; const float* coords2; char* taps; int B, h, w, S, Sh, P; };
template <int NT>
__device__ __forceinline__ void build_taps_block(const DgTapsArgs& t, const int nimg, const int cs, char* sg) {
    const int tid = threadIdx.x, HW = t.h * t.w, P = t.P, S = t.S;
    const float* coords = cs == 0 ? t.coords1 : t.coords2;
    int* cnt = reinterpret_cast<int*>(sg);                 // [HW] taps per pixel (then fill cursor)
    int* off = cnt + HW;                                   // [HW + 1] exclusive scan
    float* ewgt = reinterpret_cast<float*>(off + HW + 1);  // [4P]
    unsigned short* eidx = reinterpret_cast<unsigned short*>(ewgt + 4 * P);   // [4P]
    __shared__ int wtot[NT / 64];
    for (int i = tid; i < HW; i += NT) cnt[i] = 0;
    __syncthreads();
    int x0 = 0, y0 = 0; bool inx = false, iny = false; float w4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int pp = tid; pp < P; pp += NT) {
        const int i = pp / S, j = pp - i * S;
        dg_taps(coords + (((size_t)nimg * S + j) * t.Sh + i) * 2, t.h, t.w, x0, y0, inx, iny, w4[0], w4[1], w4[2], w4[3]);
        const int pix = y0 * t.w + x0;
        if (w4[0] != 0.f) atomicAdd(&cnt[pix], 1);
        if (inx && w4[1] != 0.f) atomicAdd(&cnt[pix + 1], 1);
        if (iny && w4[2] != 0.f) atomicAdd(&cnt[pix + t.w], 1);
        if (inx && iny && w4[3] != 0.f) atomicAdd(&cnt[pix + t.w + 1], 1);
    }
    __syncthreads();
    // exclusive scan of cnt -> off (each thread owns a contiguous run of pixels)
    const int per = (HW + NT - 1) / NT;
    const int b0 = min(tid * per, HW), b1 = min(b0 + per, HW);
    int run = 0;
    for (int i = b0; i < b1; ++i) run += cnt[i];
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
    if ((tid & 63) == 63) wtot[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int wv = 0; wv < (tid >> 6); ++wv) base += wtot[wv];
    int o2 = base + incl - run;
    for (int i = b0; i < b1; ++i) { const int c = cnt[i]; off[i] = o2; o2 += c; }
    if (tid == NT - 1) off[HW] = o2;
    __syncthreads();
    for (int i = tid; i < HW; i += NT) cnt[i] = 0;       // becomes the fill cursor
    __syncthreads();
    for (int pp = tid; pp < P; pp += NT) {
        const int i = pp / S, j = pp - i * S;
        dg_taps(coords + (((size_t)nimg * S + j) * t.Sh + i) * 2, t.h, t.w, x0, y0, inx, iny, w4[0], w4[1], w4[2], w4[3]);
        const int pix = y0 * t.w + x0;
        auto put = [&](int q, float wgt) {
            const int slot = off[q] + atomicAdd(&cnt[q], 1);
            ewgt[slot] = wgt; eidx[slot] = (unsigned short)pp;
        };
        if (w4[0] != 0.f) put(pix, w4[0]);
        if (inx && w4[1] != 0.f) put(pix + 1, w4[1]);
        if (iny && w4[2] != 0.f) put(pix + t.w, w4[2]);
        if (inx && iny && w4[3] != 0.f) put(pix + t.w + 1, w4[3]);
    }
    __syncthreads();
    // sort every pixel's list by position (insertion sort, lists are short)
    for (int q = tid; q < HW; q += NT) {
        const int e0 = off[q], e1 = off[q + 1];
        for (int i = e0 + 1; i < e1; ++i) {
            const unsigned short ki = eidx[i]; const float wi = ewgt[i];
            int j = i - 1;
            while (j >= e0 && eidx[j] > ki) { eidx[j + 1] = eidx[j]; ewgt[j + 1] = ewgt[j]; --j; }
            eidx[j + 1] = ki; ewgt[j + 1] = wi;
        }
    }
    __syncthreads();
    // copy out: [off (HW+1 ints)][weights 4P floats][positions 4P ushorts], one record per (cs, image)
    const size_t rec = dg_taps_record_bytes(HW, P);
    char* dst = t.taps + ((size_t)cs * t.B + nimg) * rec;
    int* g_off = reinterpret_cast<int*>(dst);
    float* g_w = reinterpret_cast<float*>(g_off + HW + 1);
    unsigned short* g_p = reinterpret_cast<unsigned short*>(g_w + 4 * P);
    for (int i = tid; i <= HW; i += NT) g_off[i] = off[i];
    const int ne = off[HW];
    for (int i = tid; i < ne; i += NT) { g_w[i] = ewgt[i]; g_p[i] = eidx[i]; }
}
#endif

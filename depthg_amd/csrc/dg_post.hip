// Backward tail and depth-guided sampling kernels:
//   k_scatter_grad  grid_sample backward (bilinear, border, align_corners=True; the adjoint of
//                   sample(), reference src/modules.py:822-825) of the per-pair-set code gradients,
//                   combined with the upstream gradients of the four loss means; formulated as a GATHER
//                   over an inverse tap map built in LDS (no floating-point atomics), written as (B,D,h,w) fp32.
//   k_fps_coords    farthest_point_sampling_depth (src/modules.py:999-1037) = adaptive_avg_pool2d
//                   -> depth2points(fov=90 rad, :988-996) -> fps (:939-985) -> row-major coords*2-1.
#include "dg_common.h"
#include "dg_taps.h"
#include <cstdlib>

typedef const float __attribute__((address_space(1)))* gfloat_p;    // a pointer known to be global memory


// Stage 1: comb[dest] = weighted sum of the direct sources of `dest` (gradient tiles in, gradient tile out).
// One wave = one tile of 32 sampled positions of one image and destination.  Raw sources (the fused kernel's
// accumulator tiles) are summed with their weights first, then the normalisation backward
//     dc = (dx - x <x, dx>) / max(||c||, eps)
// is applied ONCE with the operand-1 code rows (all raw sources share them); final sources (k_gs output, already
// through their own normalisation backward) are added.  grid (Ppad/32, B, 2), block 64.
// Register layout of a tile: v[d][i] = (row (i&3) + 8 (i>>2) + 4 (lane>>5), channel 32 d + (lane&31)).
// the four effective upstream gradients, loaded once (five independent loads) instead of per source
#define DG_LOAD_GS(a, gs)                                                                                   \
    float gs[4];                                                                                            \
    {                                                                                                       \
        const bool vec_ = (a).gscal != nullptr;                                                             \
        const float gt_ = vec_ ? (a).gscal[DG_OUT_TOTAL] : (a).gtot[0];                                     \
        const float g0_ = vec_ ? (a).gscal[0] : 0.f, g1_ = vec_ ? (a).gscal[1] : 0.f;                       \
        const float g2_ = vec_ ? (a).gscal[2] : 0.f, g3_ = vec_ ? (a).gscal[3] : 0.f;                       \
        gs[0] = g0_ + gt_ * (a).wtot[0]; gs[1] = g1_ + gt_ * (a).wtot[1];                                  \
        gs[2] = g2_ + gt_ * (a).wtot[2]; gs[3] = g3_ + gt_ * (a).wtot[3];                                  \
    }
__device__ __forceinline__ float dg_pick(const float (&gs)[4], int i) { return i == 0 ? gs[0] : (i == 1 ? gs[1] : (i == 2 ? gs[2] : gs[3])); }

// One block per (tile, image, destination), one WAVE PER 32-CHANNEL GROUP: a wave loads its group's slice of up to four
// sources at a time (16 independent 16-byte loads per lane) - the one-wave form walked the eight raw sources one memory latency
// after the other, twelve loads each, and was a latency chain (21 of its 26 us with every load an L2 hit).  The only thing the
// groups share is <x, dx> per position: 32 partial sums per wave through LDS, added in group order.
template <int NDF>
__global__ __launch_bounds__(64 * NDF) void k_grad_combine(const DgScatterArgs a) {
    const int rt = blockIdx.x, n = blockIdx.y, dest = blockIdx.z;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, DP = NDF * 32;
    const int d = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (dest >= 2) {       // an axpy slice (DgScatterArgs.naxpy): this block's tile of job dest - 2, one channel group per wave
        const int j = dest - 2;
        const size_t off = ((size_t)n * (a.Ppad >> 5) + rt) * (32 * DP) + d * 1024 + lane * 4;
        const float f = a.axf[j] ? a.axf[j][0] : 0.f;
        const float* const d2 = a.axd2[j];
        const float* const s1 = a.axs[j];
        const float* const s2 = a.axs2[j];
        f32x4 y[4], y2[4], x[4], x2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            y[g] = *reinterpret_cast<const f32x4*>(a.axd[j] + off + g * 256);
            y2[g] = d2 ? *reinterpret_cast<const f32x4*>(d2 + off + g * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
            x[g] = s1 ? *reinterpret_cast<const f32x4*>(s1 + off + g * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
            x2[g] = s2 ? *reinterpret_cast<const f32x4*>(s2 + off + g * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[g][e] = fmaf(f, x[g][e] + x2[g][e], y[g][e] + y2[g][e]);
            *reinterpret_cast<f32x4*>(a.axo[j] + off + g * 256) = y[g];
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) char xs[NDF][DG_XROWS_LDS];
    __shared__ float part[NDF][32];
    DG_LOAD_GS(a, gs)
    // padding channels are not stored by the producers (the bytes there are whatever the workspace held, NaN patterns
    // included): those lanes load a valid address and the value is SELECTED away, never multiplied
    const bool ok = 32 * d + r < a.D;
    const size_t tile_off = ((size_t)n * (a.Ppad >> 5) + rt) * (32 * DP) + lane * 4 + (ok ? d * 1024 : 0);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    auto add_sources = [&](const int8_t* list, const int cnt) __attribute__((always_inline)) {
        for (int k0 = 0; k0 < cnt; k0 += 4) {
            f32x4 t[4][4];
            float sc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool live = k0 + u < cnt;
                const DgScatterSrc& q = a.src[(int)list[live ? k0 + u : k0]];
                sc[u] = live ? dg_src_factor(q) * dg_pick(gs, q.gidx) : 0.f;
                const float* base = q.buf + tile_off;
#pragma unroll
                for (int g = 0; g < 4; ++g) t[u][g] = *reinterpret_cast<const f32x4*>(base + (ok ? g * 256 : 0));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool use = ok && k0 + u < cnt;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = fmaf(sc[u], use ? t[u][g][e] : 0.f, v[4 * g + e]);
            }
        }
    };
    const int nraw = a.ncraw[dest];
    add_sources(a.craw[dest], nraw);
    if (nraw > 0) {
        const char* xb = a.xop + ((size_t)n * (a.Ppad >> 5) + rt) * a.blob_bytes + a.blob_off_c + d * 2048;
        _Float16 x[1][16];
        dg_load_code_rows<1>(xb, xs[d], lane, x);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float t = half_sum((float)x[0][i] * v[i]);
            if (r == 0) part[d][h * 16 + i] = t;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t = part[0][h * 16 + i];
#pragma unroll
            for (int f = 1; f < NDF; ++f) t += part[f][h * 16 + i];
            const int rr = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const float inv = rr < a.P ? a.xinv[(size_t)n * a.Ppad + rr] : 0.f;
            v[i] = (v[i] - (float)x[0][i] * t) * inv;
        }
    }
    add_sources(a.cfin[dest], a.ncfin[dest]);
    if (ok) {
        float* out = a.comb[dest] + tile_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            *reinterpret_cast<f32x4*>(out + g * 256) = t;
        }
    }
}

// Identity grid (DG_IDENTITY_GRID; positions = pixel indices): stage 1 and the adjoint of sample() in ONE launch.  The block of
// k_grad_combine - (tile, image, destination), one wave per 32-channel group - also adds the tiles of the images ROUTED here (the
// negatives whose batch map points at this image: list built per block by ordered ballot compaction, (source, image) order:
// bit-reproducible) and writes (B,D,h,w) directly: a tile's 32 positions are 32 consecutive pixels, so each channel row of the
// tile is one 128-byte run (through a per-wave LDS stage, two channel rows per store instruction).  No combined-tile round trip,
// no second launch.
#ifndef COMB_HB_HM
#define COMB_HB_HM 4          // ... in half mode (6, 8: 176 registers, three instead of four-plus waves per SIMD: 44 us against 27.5)
#endif
#ifndef COMB_HB
#define COMB_HB 4          // fp16 tiles per batch of k_combine_out (two 16-byte loads each; 8: 176 registers, two waves per SIMD, 48 us against 34)
#endif
#define COMB_MAXROUTE 512      // (routed source, image) pairs a destination image can collect at worst: routed sources x B
// HM (half mode: the call's pair-set tiles are fp16, only the depth term's are fp32): eight fp16 tiles and ONE fp32 tile per batch instead
// of four and four - the block is a chain of load phases (ablations, experiments/r06.md 18.8: each costs 3-5 us of the launch), and
// with every raw tile of a destination in one batch and every routed one in another there are four of them instead of seven
template <int NDF, bool HM = false>
__global__ __launch_bounds__(64 * NDF) void k_combine_out(const DgScatterArgs a) {
    const int rt = blockIdx.x, n = blockIdx.y, dest = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, DP = NDF * 32;
    const int d = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ __attribute__((aligned(16))) char xs[NDF][DG_XROWS_LDS];
    __shared__ float part[NDF][32];
    __shared__ float stage[NDF][32 * 33];
    __shared__ const float* rl_p[COMB_MAXROUTE];
    __shared__ float rl_w[COMB_MAXROUTE];
    __shared__ int rl_cnt, wcnt[NDF], rsrc[DG_MAX_SCATTER];
    DG_LOAD_GS(a, gs)
    // ---- routed list of (dest, n): every thread tests one (routed source, image) pair per round
    int nr = 0;
    if (DG_DBG(a.debug) & 8) { if (dest == 1) return; }                   // (developer ablations, WRONG results: 8 no destination-1 blocks,
    for (int s = 0; s < a.nsrc; ++s)                                       //  1 no routed sources, 2 no norm() backward, 4 no output)
        if (a.src[s].dest == dest && a.src[s].route != nullptr && !(DG_DBG(a.debug) & 1)) { if (tid == 0) rsrc[nr] = s; ++nr; }
    if (tid == 0) rl_cnt = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nr * a.B; i0 += 64 * NDF) {
        const int i = i0 + tid;
        bool hit = false;
        int m = 0, s = 0;
        if (i < nr * a.B) {
            const int rs = i / a.B;
            m = i - rs * a.B; s = rsrc[rs];
            hit = (int)a.src[s].route[m] == n;
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcnt[d] = __popcll(bal);
        __syncthreads();
        int base = rl_cnt;
        for (int wv = 0; wv < d; ++wv) base += wcnt[wv];
        if (hit) {
            const int o = base + __popcll(bal & ((1ull << lane) - 1));
            const DgScatterSrc& q = a.src[s];
            if (o < COMB_MAXROUTE) {
                // (half tiles: the same element index, two bytes per element - the pointer stays a float* and is re-typed at the load)
                rl_p[o] = q.half ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(q.buf) + (size_t)m * a.Ppad * a.DP) : q.buf + (size_t)m * a.Ppad * a.DP;
                rl_w[o] = dg_src_factor(q) * dg_pick(gs, q.gidx);
            }
        }
        __syncthreads();
        if (tid == 0) {
            int c = rl_cnt;
            for (int wv = 0; wv < NDF; ++wv) c += wcnt[wv];
            rl_cnt = min(c, COMB_MAXROUTE);
        }
        __syncthreads();
    }
    const int cnt = rl_cnt;
    // ---- the direct sources, as k_grad_combine
    const bool ok = 32 * d + r < a.D;
    const size_t in_img = (size_t)rt * (32 * DP) + lane * 4 + (ok ? d * 1024 : 0);
    const size_t tile_off = (size_t)n * a.Ppad * DP + in_img;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    // fp16 tiles (DgScatterSrc.half): [2][64][8] per channel group - the lane's elements 8s .. 8s+7 in one 16-byte piece; final ones are
    // projected but not yet divided by ||c||: 1 / ||c|| of the destination's own positions, loaded once
    const size_t in_img_h = (size_t)rt * (32 * DP) + lane * 8 + (ok ? d * 1024 : 0);
    constexpr int HB = HM ? COMB_HB_HM : COMB_HB, FB = HM ? 1 : 4;
    float invd[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        invd[i] = (a.xinv_dest[dest] && rr < a.P) ? a.xinv_dest[dest][(size_t)n * a.Ppad + rr] : 0.f;
    }
    const _Float16 __attribute__((address_space(1)))* const hdummy = reinterpret_cast<const _Float16 __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(a.xinv));
    // up to eight half tiles at a time (HB: sixteen 16-byte loads in flight, what four fp32 tiles are), every load of the batch in flight before the first is used (one at a time the launch was
    // bound by the latency of its loads: the fp32 form's 28 us came back as 24 instead of 17)
    auto add_half4 = [&](const _Float16 __attribute__((address_space(1)))* const (&hb)[HB], const float (&sc)[HB], const bool raw) __attribute__((always_inline)) {
        f16x8 t0[HB], t1[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            t0[u] = *reinterpret_cast<const f16x8 __attribute__((address_space(1)))*>(hb[u]);
            t1[u] = *reinterpret_cast<const f16x8 __attribute__((address_space(1)))*>(hb[u] + ((ok && hb[u] != hdummy) ? 512 : 0));
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const bool use = ok && sc[u] != 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = fmaf(raw ? sc[u] : sc[u] * invd[e], use ? (float)t0[u][e] : 0.f, v[e]);
                v[8 + e] = fmaf(raw ? sc[u] : sc[u] * invd[8 + e], use ? (float)t1[u][e] : 0.f, v[8 + e]);
            }
        }
    };
    auto add_sources_half = [&](const int8_t* list, const int nsrc_, const bool raw) __attribute__((always_inline)) {
        for (int k0 = 0; k0 < nsrc_; k0 += HB) {
            const _Float16 __attribute__((address_space(1)))* hb[HB];
            float sc[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const bool live = k0 + u < nsrc_;
                const DgScatterSrc& q = a.src[(int)list[live ? k0 + u : k0]];
                sc[u] = live ? dg_src_factor(q) * dg_pick(gs, q.gidx) : 0.f;
                // (the address does not wait for the weight - the upstream gradients are a load of their own: a tile whose weight turns
                //  out zero is fetched and dropped by the select in add_half4, never multiplied: it may be unwritten)
                hb[u] = live ? reinterpret_cast<const _Float16 __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(q.buf)) + (size_t)n * a.Ppad * DP + in_img_h
                             : hdummy;
            }
            add_half4(hb, sc, raw);
        }
    };
    auto add_sources = [&](const int8_t* list, const int nsrc_) __attribute__((always_inline)) {

        for (int k0 = 0; k0 < nsrc_; k0 += FB) {
            f32x4 t[FB][4];
            float sc[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const bool live = k0 + u < nsrc_;
                const DgScatterSrc& q = a.src[(int)list[live ? k0 + u : k0]];
                sc[u] = live ? dg_src_factor(q) * dg_pick(gs, q.gidx) : 0.f;
                const float* base = q.buf + tile_off;
#pragma unroll
                for (int g = 0; g < 4; ++g) t[u][g] = *reinterpret_cast<const f32x4*>(base + (ok ? g * 256 : 0));
            }
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const bool use = ok && k0 + u < nsrc_;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = fmaf(sc[u], use ? t[u][g][e] : 0.f, v[4 * g + e]);
            }
        }
    };
    const int nraw = a.ncraw[dest] + a.ncrawh[dest];
    add_sources(a.craw[dest], a.ncraw[dest]);
    add_sources_half(a.crawh[dest], a.ncrawh[dest], true);
    if (nraw > 0 && !(DG_DBG(a.debug) & 2)) {
        const char* xb = a.xop + ((size_t)n * (a.Ppad >> 5) + rt) * a.blob_bytes + a.blob_off_c + d * 2048;
        _Float16 x[1][16];
        dg_load_code_rows<1>(xb, xs[d], lane, x);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float t = half_sum((float)x[0][i] * v[i]);
            if (r == 0) part[d][h * 16 + i] = t;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t = part[0][h * 16 + i];
#pragma unroll
            for (int f = 1; f < NDF; ++f) t += part[f][h * 16 + i];
            const int rr = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const float inv = rr < a.P ? a.xinv[(size_t)n * a.Ppad + rr] : 0.f;
            v[i] = (v[i] - (float)x[0][i] * t) * inv;
        }
    }
    add_sources(a.cfin[dest], a.ncfin[dest]);
    add_sources_half(a.cfinh[dest], a.ncfinh[dest], false);
    // ---- the routed images' tiles (k_gs output: final), four at a time
    if (a.routed_half) {
        for (int e0 = 0; e0 < cnt; e0 += HB) {
            const _Float16 __attribute__((address_space(1)))* hb[HB];
            float sc[HB];
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int e = min(e0 + k, cnt - 1);
                sc[k] = e0 + k < cnt ? rl_w[e] : 0.f;
                hb[k] = sc[k] != 0.f ? reinterpret_cast<const _Float16 __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(rl_p[e])) + in_img_h : hdummy;
            }
            add_half4(hb, sc, false);
        }
    } else
    for (int e0 = 0; e0 < cnt; e0 += 4) {
        f32x4 u4[4][4];
        float sc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = min(e0 + k, cnt - 1);
            sc[k] = e0 + k < cnt ? rl_w[e] : 0.f;
            const gfloat_p sb = (gfloat_p)rl_p[e] + in_img;       // (pointer out of LDS: name its address space - else FLAT loads)
#pragma unroll
            for (int g = 0; g < 4; ++g) u4[k][g] = *(const f32x4 __attribute__((address_space(1)))*)(sb + (ok ? g * 256 : 0));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = fmaf(sc[k], ok ? u4[k][g][e] : 0.f, v[4 * g + e]);
    }
    // ---- out[dest][(n, channel, pixel)]: the wave's [32 channels][32 positions] tile through LDS, rows of 128 contiguous bytes
    if (DG_DBG(a.debug) & 4) { if (v[0] == 1.2345f) a.out[dest][0] = v[1]; return; }
    float* st = stage[d];
#pragma unroll
    for (int i = 0; i < 16; ++i) st[r * 33 + (i & 3) + 8 * (i >> 2) + 4 * h] = v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int HW = a.h * a.w, p = rt * 32 + r;
    float* out = a.out[dest] + (size_t)n * a.D * HW + p;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int c = 2 * it + h, ch = 32 * d + c;
        if (ch < a.D && p < a.P) out[(size_t)ch * HW] = st[c * 33 + r];
    }
}

#define SCAT_THREADS 1024
#define SCAT_DC 4            // channels per block (4 x 256 pixels per pass: the gather keeps 4 sources x 4 passes of loads in flight in 48 registers)
#define SCAT_PX (SCAT_THREADS / SCAT_DC)
#define SCAT_MAXPASS 16      // supports h*w <= SCAT_PX * SCAT_MAXPASS = 4096 pixels

__global__ __launch_bounds__(SCAT_THREADS) void k_build_taps(const DgScatterArgs a) {
    extern __shared__ __attribute__((aligned(16))) char sg[];
    const DgTapsArgs t{a.coords1, a.coords2, a.taps, a.B, a.h, a.w, a.S, a.Sh, a.P};
    build_taps_block<SCAT_THREADS>(t, (int)blockIdx.x, (int)blockIdx.y, sg);
}

__global__ __launch_bounds__(256) void k_pre_general(const DgPreArgs a) {
    extern __shared__ __attribute__((aligned(16))) char pg[];
    int b = (int)blockIdx.x;
    if (a.zero_word && b == 0 && threadIdx.x == 0) *a.zero_word = 0u;
    if (b < a.count) { dg_super_perm_row(nullptr, a.seed, a.state, a.B, a.perms, b, a.count, reinterpret_cast<float*>(pg)); return; }
    b -= a.count;
    if (a.depth) {
        if (b < a.B) { depth_nz_image(a.depth, a.nz, a.nzsum, b, a.dH, a.dW, a.Sh, a.S, a.Ppad); return; }
        b -= a.B;
    }
    if (!a.taps) return;                       // (a launch that only zeroes the ticket word)
    const DgTapsArgs t{a.coords1, a.coords2, a.taps, a.B, a.h, a.w, a.S, a.Sh, a.P};
    build_taps_block<256>(t, b % a.B, b / a.B, pg);
}

hipError_t dg_launch_pre_general(const DgPreArgs& a, hipStream_t s) {
    int nblk = a.count + (a.depth ? a.B : 0) + (a.taps ? 2 * a.B : 0);
    if (nblk == 0 && a.zero_word) nblk = 1;
    if (nblk == 0) return hipSuccess;
    size_t smem = a.count > 0 ? (size_t)a.B * 4 : 0;
    if (a.taps) {
        const size_t need = dg_taps_record_bytes(a.h * a.w, a.P) + (size_t)a.h * a.w * 4 + 16;
        smem = need > smem ? need : smem;
    }
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(k_pre_general), (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pre_general, dim3(nblk), dim3(256), smem, s, a);
    return hipGetLastError();
}

// Stage 2: adjoint of sample() as a GATHER (no floating-point atomics).  One block = one destination image and
// SCAT_DC channels.  The sources that land there - the combined direct tiles, then the negatives routed here, in (source,
// image) order - are taken SCAT_RMAX at a time: their tap lists go to LDS together (one global latency and one barrier per
// batch instead of per source), then every thread gathers its (pixel, channel) outputs into registers, source after source
// in the fixed order, taps in position order: rows of the fp32 gradient buffers are read, nothing is scattered, the sums
// are bit-reproducible.
// grid (ceil(D / SCAT_DC), B, 2), block SCAT_THREADS, dynamic LDS = rmax tap records (>= the output staging tile).
#define SCAT_RMAX 8
template <int NPASS>
__global__ __launch_bounds__(SCAT_THREADS) void k_scatter_grad(const DgScatterArgs a, int rmax) {
    extern __shared__ __attribute__((aligned(16))) char sg[];
    const int tid = threadIdx.x, HW = a.h * a.w, P = a.P;
    // (the lambdas below capture these copies, never `a` itself: a by-reference capture of the kernel argument makes hipcc
    //  copy the whole 1.5-KB struct to scratch in every thread)
    const int aD = a.D, aDP = a.DP, aB = a.B, aPpad = a.Ppad;
    const char* const taps_base = a.taps;
    DG_LOAD_GS(a, gs)
    const int dc0 = blockIdx.x * SCAT_DC, bdst = blockIdx.y, dest = blockIdx.z;
    const int dl = tid & (SCAT_DC - 1), px = tid / SCAT_DC, d = dc0 + dl;
    const int npass = (HW + SCAT_PX - 1) / SCAT_PX;
    const size_t rec = dg_taps_record_bytes(HW, P);
    // the pending batch (uniform over the block; thread 0 writes, everybody reads behind the batch's barrier)
    __shared__ const float* b_src[SCAT_RMAX];
    __shared__ const char* b_rec[SCAT_RMAX];
    __shared__ float b_sc[SCAT_RMAX];
    float acc[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) acc[i] = 0.f;
    int npend = 0;

    auto flush = [&, aD, aDP]() __attribute__((always_inline)) {
        if (npend == 0) return;
        __syncthreads();                                   // batch entries written
        const int n16 = (int)(rec / 16);
        for (int i = tid; i < npend * n16; i += SCAT_THREADS) {
            const int r = i / n16, k = i - r * n16;
            reinterpret_cast<f32x4*>(sg)[i] = *((const f32x4 __attribute__((address_space(1)))*)b_rec[r] + k);
        }
        __syncthreads();
        if (d < aD) {
            const int DPc = aDP;
            // RG sources at a time: the first tap of every (source, pixel) of this thread is loaded before anything is summed -
            // RG * NPASS independent loads in flight instead of one dependent load chain per source (most pixels are read by at
            // most one position of a source); further taps of a pixel, rare, follow in the summation pass
            constexpr int RG = NPASS <= 4 ? 4 : (NPASS <= 8 ? 2 : 1);
            for (int r0 = 0; r0 < npend; r0 += RG) {
                float v[RG][NPASS], w[RG][NPASS];
                int e0[RG][NPASS];                         // first tap; bit 31: the pixel has further taps in this source
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    const int r = min(r0 + g, npend - 1);
                    const int* off = reinterpret_cast<const int*>(sg + (size_t)r * rec);
                    const float* ewgt = reinterpret_cast<const float*>(off + HW + 1);
                    const unsigned short* eidx = reinterpret_cast<const unsigned short*>(ewgt + 4 * P);
                    // (the pointer comes out of LDS: without the address space hipcc emits FLAT loads, which also count as LDS
                    //  operations - every later ds_read then waits for all of them: 10x slower)
                    const gfloat_p src = (gfloat_p)b_src[r];
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) {
                        const int pix = ps * SCAT_PX + px;
                        const bool ok = r0 + g < npend && ps < npass && pix < HW;
                        const int ea = ok ? off[pix] : 0, eb = ok ? off[pix + 1] : 0;
                        const bool any = eb > ea;
                        w[g][ps] = any ? ewgt[ea] : 0.f;
                        v[g][ps] = any ? src[dg_gtile_off(eidx[ea], d, DPc)] : 0.f;
                        e0[g][ps] = ea | (eb > ea + 1 ? (int)0x80000000 : 0);
                    }
                }
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    if (r0 + g < npend) {
                        const int r = r0 + g;
                        const int* off = reinterpret_cast<const int*>(sg + (size_t)r * rec);
                        const float* ewgt = reinterpret_cast<const float*>(off + HW + 1);
                        const unsigned short* eidx = reinterpret_cast<const unsigned short*>(ewgt + 4 * P);
                        const gfloat_p src = (gfloat_p)b_src[r];
                        const float sc = b_sc[r];
#pragma unroll
                        for (int ps = 0; ps < NPASS; ++ps) {
                            float sum = w[g][ps] * v[g][ps];
                            if (e0[g][ps] < 0) {
                                const int ea = e0[g][ps] & 0x7fffffff, eb = off[ps * SCAT_PX + px + 1];
                                for (int e = ea + 1; e < eb; ++e) sum = fmaf(ewgt[e], src[dg_gtile_off(eidx[e], d, DPc)], sum);
                            }
                            acc[ps] = fmaf(sc, sum, acc[ps]);
                        }
                    }
                }
            }
        }
        __syncthreads();                                   // everybody is done with this batch's entries and tap lists
        npend = 0;
    };
    auto push = [&](const float* buf, float sc, int cs, int nimg) __attribute__((always_inline)) {
        if (tid == 0) {
            b_src[npend] = buf + (size_t)nimg * aPpad * aDP;              // gradient tiles of image nimg
            b_rec[npend] = taps_base + ((size_t)cs * aB + nimg) * rec;    // its tap lists (coords set cs)
            b_sc[npend] = sc;
        }
        if (++npend == rmax) flush();
    };

    // direct sources (already combined): image bdst -> destination bdst; coords1 for grad_code, coords2 for grad_code_pos
    push(a.comb[dest], 1.0f, dest == 0 ? 0 : 1, bdst);
    // routed sources (negatives): image n scatters into destination route[n] with n's coords
    for (int s = 0; s < a.nsrc; ++s) {
        const DgScatterSrc& q = a.src[s];
        if (q.dest != dest || q.route == nullptr) continue;
        const float sc = dg_src_factor(q) * dg_pick(gs, q.gidx);
        for (int n0 = 0; n0 < a.B; n0 += 64) {           // which images route here: one ballot per 64 images
            const int nn = n0 + (tid & 63);
            const bool hit = nn < a.B && (int)q.route[nn] == bdst;
            unsigned long long m = __ballot(hit);
            while (m) {
                const int n = n0 + __builtin_ctzll(m);
                m &= m - 1;
                push(q.buf, sc, q.coords_sel, n);
            }
        }
    }
    flush();
    // ---- write (B,D,h,w): transpose through LDS so that the stores run along the pixels
    float* stage = reinterpret_cast<float*>(sg);     // [SCAT_DC][HW + 1]
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int pix = ps * SCAT_PX + px;
        if (ps < npass && pix < HW) stage[dl * (HW + 1) + pix] = acc[ps];
    }
    __syncthreads();
    float* out = a.out[dest];
    for (int i = tid; i < SCAT_DC * HW; i += SCAT_THREADS) {
        const int dd = i / HW, pix = i - dd * HW;
        if (dc0 + dd < a.D) out[((size_t)bdst * a.D + dc0 + dd) * HW + pix] = stage[dd * (HW + 1) + pix];
    }
}

// Stage 2 for small sample grids (the S = 11 / 12 recipes: P = 121 / 144 positions on a 28 x 28 map), same sums in the same
// order as k_scatter_grad, different mapping: one block = one destination image and one group of 32 channels.  The gradient
// tiles of that group of up to `rb` sources are copied into LDS as [position][33] rows (coalesced 16-byte loads of whole
// tiles; every byte of a tile is read once per block instead of one 64-byte line per tap and four channels), next to the
// sources' tap lists.  Then the LANES RUN OVER PIXELS: a thread walks the tap lists of its pixel once for all 32 channels and
// keeps 32 sums in registers; the result is stored along the pixels (coalesced, no transposition stage).
// grid (DP / 32, B, 2), block SCAT_THREADS (needs h*w <= SCAT_THREADS), dynamic LDS rb * (record + Ppad * 33 floats).
template <int CG>        // channels per block: 32, or 16 (twice the blocks, half the LDS) when 32 would leave CUs idle
__global__ __launch_bounds__(SCAT_THREADS) void k_scatter_small(const DgScatterArgs a, int rb) {
    constexpr int VS = CG + 1;                              // padded row of the value image
    extern __shared__ __attribute__((aligned(16))) char sg[];
    const int tid = threadIdx.x, HW = a.h * a.w, P = a.P, nt = a.Ppad >> 5, NF = a.DP >> 5;
    const int aB = a.B, aPpad = a.Ppad, aDP = a.DP;
    const char* const taps_base = a.taps;
    DG_LOAD_GS(a, gs)
    const int f = blockIdx.x / (32 / CG), cg = blockIdx.x % (32 / CG), bdst = blockIdx.y, dest = blockIdx.z;
    const size_t rec = dg_taps_record_bytes(HW, P);
    const int vstride = aPpad * VS;                         // floats per source
    float* const vals = reinterpret_cast<float*>(sg + (size_t)rb * rec);
    __shared__ const float* b_src[SCAT_RMAX];
    __shared__ const char* b_rec[SCAT_RMAX];
    __shared__ float b_sc[SCAT_RMAX];
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = 0.f;
    int npend = 0;

    auto flush = [&, rec, nt, NF, f, cg, vstride, HW, P]() __attribute__((always_inline)) {
        if (npend == 0) return;
        __syncthreads();                                   // batch entries written
        const int n16 = (int)(rec / 16);
        for (int i = tid; i < npend * n16; i += SCAT_THREADS) {
            const int r = i / n16, k = i - r * n16;
            reinterpret_cast<f32x4*>(sg)[i] = *((const f32x4 __attribute__((address_space(1)))*)b_rec[r] + k);
        }
        // gradient tiles of channel group f: 256 16-byte pieces per tile = {4 consecutive positions of one channel}
        constexpr int PT = 8 * CG;                          // pieces of this block's channels per tile
        for (int i = tid; i < npend * nt * PT; i += SCAT_THREADS) {
            const int r = i / (nt * PT), j = i - r * (nt * PT), t = j / PT, k = j - t * PT;
            const int c = k % CG, hq = k / CG;               // hq = 2 (q >> 3) + ((q >> 2) & 1): which four positions
            const int jj = (hq >> 1) * 64 + (hq & 1) * 32 + CG * cg + c;
            const f32x4 v = *((const f32x4 __attribute__((address_space(1)))*)(b_src[r] + ((size_t)t * NF + f) * 1024) + jj);
            float* o = vals + (size_t)r * vstride + (t * 32 + hq * 4) * VS + c;
            o[0] = v[0]; o[VS] = v[1]; o[2 * VS] = v[2]; o[3 * VS] = v[3];
        }
        __syncthreads();
        if (tid < HW) {
            for (int r = 0; r < npend; ++r) {
                const int* off = reinterpret_cast<const int*>(sg + (size_t)r * rec);
                const float* ewgt = reinterpret_cast<const float*>(off + HW + 1);
                const unsigned short* eidx = reinterpret_cast<const unsigned short*>(ewgt + 4 * P);
                const float* vr = vals + (size_t)r * vstride;
                const int e0 = off[tid], e1 = off[tid + 1];
                const float sc = b_sc[r];
                for (int e = e0; e < e1; ++e) {            // (weight of the source folded into the tap weight: one fma per tap and channel)
                    const float w = sc * ewgt[e];
                    const float* v = vr + (int)eidx[e] * VS;
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc[c] = fmaf(w, v[c], acc[c]);
                }
            }
        }
        __syncthreads();                                   // everybody is done with this batch
        npend = 0;
    };
    auto push = [&](const float* buf, float sc, int cs, int nimg) __attribute__((always_inline)) {
        if (tid == 0) {
            b_src[npend] = buf + (size_t)nimg * aPpad * aDP;
            b_rec[npend] = taps_base + ((size_t)cs * aB + nimg) * rec;
            b_sc[npend] = sc;
        }
        if (++npend == rb) flush();
    };
    // The sources in a fixed order: the combined direct tiles, then the routed ones (negatives) by (image chunk, source, image).
    // Per 64 images and 16 sources the route entries are loaded together (independent loads: one memory latency instead of
    // one per source); the hits are listed in LDS by thread 0 and pushed from ONE call site (the batch logic is inlined once).
    __shared__ short l_s[16 * 64 + 1], l_n[16 * 64 + 1];
    for (int n0 = 0; n0 < a.B; n0 += 64) {
        const int nn = n0 + (tid & 63);
        for (int s0 = 0; s0 < a.nsrc || s0 == 0; s0 += 16) {
            int cnt = 0;
            if (n0 == 0 && s0 == 0) { if (tid == 0) { l_s[0] = -1; l_n[0] = (short)bdst; } cnt = 1; }
            int rt[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int s = s0 + k;
                const bool use = s < a.nsrc && a.src[s < a.nsrc ? s : 0].dest == dest && a.src[s < a.nsrc ? s : 0].route != nullptr && nn < a.B;
                rt[k] = use ? (int)a.src[s].route[nn] : -1;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned long long m = __ballot(rt[k] == bdst);
                if (tid == 0) {
                    int c = cnt;
                    for (unsigned long long mm = m; mm; mm &= mm - 1) { l_s[c] = (short)(s0 + k); l_n[c] = (short)(n0 + __builtin_ctzll(mm)); ++c; }
                }
                cnt += __popcll(m);
            }
            __syncthreads();
            for (int i = 0; i < cnt; ++i) {
                const int s = l_s[i], n = l_n[i];
                const DgScatterSrc& q = a.src[s < 0 ? 0 : s];
                const float* buf = s < 0 ? a.comb[dest] : q.buf;
                const float sc = s < 0 ? 1.0f : dg_src_factor(q) * dg_pick(gs, q.gidx);
                const int cs = s < 0 ? (dest == 0 ? 0 : 1) : q.coords_sel;
                push(buf, sc, cs, n);
            }
            __syncthreads();
        }
    }
    flush();
    if (tid < HW) {
        const int d0 = 32 * f + CG * cg;
        float* out = a.out[dest] + ((size_t)bdst * a.D + d0) * HW + tid;
#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (d0 + c < a.D) out[(size_t)c * HW] = acc[c];
    }
}

// Stage 2 on the identity grid (DG_IDENTITY_GRID): sample() reads pixel (y = j, x = i) for position p = i*S + j with
// weight 1, so its adjoint is a transposed copy.  One block = one destination image, one group of 32 channels: every
// wave sums, tile by tile, the combined direct gradient tile and the tiles of the images routed here (negatives,
// in a fixed order: bit-reproducible), drops the rows into an LDS stage [channel][pixel] and the block writes (B,D,h,w)
// rows of h*w contiguous floats.  grid (DP/32, B, 2), block 1024, dynamic LDS 32*(HW+1) floats + the routed list.
#define DENSE_MAXROUTE 1024
// CG channels per block (32, 16 or 8): a wave covers 32 / CG tiles at once - lanes [2 CG ts, 2 CG (ts + 1)) take tile ts of the
// group, as the lanes 32 h + CG cg + idx of the gradient-tile layout (runs of CG lanes = CG * 16 contiguous bytes).  Smaller CG =
// more, lighter blocks: the destination that collects the negatives' gradients is several times heavier than the other one,
// and with 32-channel blocks only 192 blocks (96 heavy) cover the 256 CUs.
template <int CG, int DENSE_THREADS>
__global__ __launch_bounds__(DENSE_THREADS) void k_scatter_dense(const DgScatterArgs a) {
    extern __shared__ __attribute__((aligned(16))) char sd[];
    const int HW = a.h * a.w, nt = a.Ppad >> 5, NF = a.DP >> 5;
    DG_LOAD_GS(a, gs)
    float* stage = reinterpret_cast<float*>(sd);                       // [CG][HW + 1]
    const float** rl_p = reinterpret_cast<const float**>(stage + CG * (HW + 2));   // routed list: image base (8-byte aligned)
    float* rl_w = reinterpret_cast<float*>(rl_p + DENSE_MAXROUTE);                  //              weight
    __shared__ int rl_cnt;
    constexpr int TPW = 32 / CG;                                       // tiles per wave and round
    const int f = blockIdx.x / TPW, cg = blockIdx.x % TPW, b = blockIdx.y, dest = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ts = lane / (2 * CG), h = (lane / CG) & 1, idx = lane % CG;
    const int r = CG * cg + idx;                                       // channel inside the 32-channel group = lane of the layout
    // routed list in (source, image) order: the (routed source, image) pairs are numbered source-major, every thread tests
    // one pair per round (all sources at once), ballot + prefix compaction keeps the order
    constexpr int NW = DENSE_THREADS / 64;
    __shared__ int wcnt[NW];
    __shared__ int rsrc[DG_MAX_SCATTER];
    int nr = 0;
    for (int s = 0; s < a.nsrc; ++s)                                   // uniform: which sources are routed here
        if (a.src[s].dest == dest && a.src[s].route != nullptr) { if (tid == 0) rsrc[nr] = s; ++nr; }
    if (tid == 0) rl_cnt = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nr * a.B; i0 += DENSE_THREADS) {
        const int i = i0 + tid;
        bool hit = false;
        int n = 0, s = 0;
        if (i < nr * a.B) {
            const int rs = i / a.B;
            n = i - rs * a.B; s = rsrc[rs];
            hit = (int)a.src[s].route[n] == b;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wcnt[wid] = __popcll(m);
        __syncthreads();
        int base = rl_cnt;
        for (int wv = 0; wv < wid; ++wv) base += wcnt[wv];
        if (hit) {
            const int o = base + __popcll(m & ((1ull << lane) - 1));
            const DgScatterSrc& q = a.src[s];
            if (o < DENSE_MAXROUTE) { rl_p[o] = q.buf + (size_t)n * a.Ppad * a.DP; rl_w[o] = dg_src_factor(q) * dg_pick(gs, q.gidx); }
        }
        __syncthreads();
        if (tid == 0) {
            int c = rl_cnt;
            for (int wv = 0; wv < NW; ++wv) c += wcnt[wv];
            rl_cnt = min(c, DENSE_MAXROUTE);
        }
        __syncthreads();
    }
    const int cnt = rl_cnt;
    for (int t0 = wid * TPW; t0 < nt; t0 += NW * TPW) {
        const int t = t0 + ts;
        const bool live = t < nt;
        const size_t toff = ((size_t)(live ? t : 0) * NF + f) * 1024 + (32 * h + r) * 4;
        f32x4 v[4];
        const float* cb = a.comb[dest] + (size_t)b * a.Ppad * a.DP + toff;
        const bool chan = live && 32 * f + r < a.D;           // padding channels are not stored by the producers
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = chan ? *reinterpret_cast<const f32x4*>(cb + g * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int e0 = 0; e0 < cnt; e0 += 4) {                 // 4 routed images per round: 16 loads in flight
            f32x4 u[4][4];
            float sc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = min(e0 + k, cnt - 1);
                sc[k] = e0 + k < cnt ? rl_w[e] : 0.f;
                // (pointer out of LDS: name its address space, or hipcc emits FLAT loads, which every later LDS read waits for)
                const gfloat_p sb = (gfloat_p)rl_p[e] + toff;
#pragma unroll
                for (int g = 0; g < 4; ++g) u[k][g] = chan ? *(const f32x4 __attribute__((address_space(1)))*)(sb + g * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[g][c] = fmaf(sc[k], u[k][g][c], v[g][c]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = t * 32 + k + 8 * g + 4 * h;
                if (live && p < a.P) stage[idx * (HW + 1) + p] = v[g][k];      // position = pixel index on the identity grid
            }
    }
    __syncthreads();
    float* out = a.out[dest];
    for (int c = 0; c < CG; ++c) {
        const int d = 32 * f + CG * cg + c;
        if (d >= a.D) break;
        for (int pix = tid; pix < HW; pix += DENSE_THREADS) out[((size_t)b * a.D + d) * HW + pix] = stage[c * (HW + 1) + pix];
    }
}

hipError_t dg_launch_scatter(const DgScatterArgs& a, hipStream_t s) {
    DgScatterArgs ac = a;
    {
        for (int dest = 0; dest < 2; ++dest) {
            ac.ncraw[dest] = ac.ncfin[dest] = ac.ncrawh[dest] = ac.ncfinh[dest] = 0;
            for (int i = 0; i < a.nsrc; ++i) {
                const DgScatterSrc& q = a.src[i];
                if (q.dest != dest || q.route != nullptr) continue;
                int8_t& cnt = q.half ? (q.raw ? ac.ncrawh[dest] : ac.ncfinh[dest]) : (q.raw ? ac.ncraw[dest] : ac.ncfin[dest]);
                if (cnt >= DG_MAX_SCATTER / 2) return hipErrorInvalidValue;
                (q.half ? (q.raw ? ac.crawh[dest] : ac.cfinh[dest]) : (q.raw ? ac.craw[dest] : ac.cfin[dest]))[cnt++] = (int8_t)i;
            }
        }
        const dim3 cgrid(a.Ppad / 32, a.B, 2);
#ifndef DG_TWO_STAGE_DENSE      // (developer A/B: the round-2 form, k_grad_combine + k_scatter_dense)
        {
            // identity grid: combine + routed negatives + (B,D,h,w) output in one launch
            int nrouted = 0;
            for (int i = 0; i < a.nsrc; ++i) nrouted += a.src[i].route != nullptr;
            bool any_half = false;
            int routed_h = 0;
            for (int i = 0; i < a.nsrc; ++i) { any_half = any_half || a.src[i].half != 0; routed_h += (a.src[i].route != nullptr && a.src[i].half) ? 1 : 0; }
            if (routed_h != 0 && routed_h != nrouted) return hipErrorInvalidValue;      // (fp16 tiles for every routed source or for none)
            ac.routed_half = routed_h != 0 ? 1 : 0;
#ifdef DG_DEVTOOLS
            if (const char* e = getenv("DG_COMB_DEBUG")) ac.debug = atoi(e);
#endif
            const bool one_launch = a.dense && a.S == a.h && a.S == a.w && nrouted * a.B <= COMB_MAXROUTE && (a.DP == 96 || a.DP == 128);
            if (any_half && !one_launch) return hipErrorInvalidValue;      // (fp16 tiles are k_combine_out's: the plan asks for them only where it runs)
            if (one_launch) {
                // (the list holds the worst case - every routed image of every routed source pointing at one destination)
                if (a.DP == 96 && any_half) hipLaunchKernelGGL((k_combine_out<3, true>), cgrid, dim3(192), 0, s, ac);
                else if (a.DP == 96) hipLaunchKernelGGL(k_combine_out<3>, cgrid, dim3(192), 0, s, ac);
                else hipLaunchKernelGGL(k_combine_out<4>, cgrid, dim3(256), 0, s, ac);
                return hipGetLastError();
            }
        }
#endif
        const dim3 cgrid2(a.Ppad / 32, a.B, 2 + a.naxpy);
        if (a.DP == 96) hipLaunchKernelGGL(k_grad_combine<3>, cgrid2, dim3(192), 0, s, ac);
        else if (a.DP == 128) hipLaunchKernelGGL(k_grad_combine<4>, cgrid2, dim3(256), 0, s, ac);
        else return hipErrorInvalidValue;
    }
    const int HW = a.h * a.w;
    {
        // identity grid: transposed copy (needs the [32][HW+1] stage in LDS and a bounded routed list)
        int nrouted = 0;
        for (int i = 0; i < a.nsrc; ++i) nrouted += a.src[i].route != nullptr;
        // 16 channels per block: 27 -> 21 us at the headline shape (8: 22 us); DG_SCATTER_CG overrides (developer A/B)
#ifdef DG_DEVTOOLS
        static const int cgsel = getenv("DG_SCATTER_CG") ? atoi(getenv("DG_SCATTER_CG")) : 16;
#else
        constexpr int cgsel = 16;
#endif
        int CGv = cgsel == 32 ? 32 : (cgsel == 8 ? 8 : 16);
        auto stage_bytes = [&](int cg) { return (size_t)cg * (HW + 2) * 4 + (size_t)DENSE_MAXROUTE * 12; };
        if (stage_bytes(CGv) > 150 * 1024 && CGv > 8) CGv = 8;      // 56 x 56 maps: 8 channels per block keep the [channel][pixel] stage in LDS
        const size_t dsm = stage_bytes(CGv);
        if (a.dense && a.S == a.h && a.S == a.w && dsm <= 150 * 1024 && (size_t)nrouted * a.B <= DENSE_MAXROUTE) {
            auto launch = [&](auto kern, int threads) -> hipError_t {
                hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), (int)dsm);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(a.DP / CGv, a.B, 2), dim3(threads), dsm, s, a);
                return hipGetLastError();
            };
            if (CGv == 32) return launch(k_scatter_dense<32, 1024>, 1024);
            if (CGv == 16) return launch(k_scatter_dense<16, 512>, 512);
            return launch(k_scatter_dense<8, 512>, 512);
        }
    }
    if (HW > SCAT_PX * SCAT_MAXPASS || a.P > 65535) return hipErrorInvalidValue;
    const size_t rec = dg_taps_record_bytes(HW, a.P);
    const size_t stage = (size_t)SCAT_DC * (HW + 1) * 4;
    int rmax = (int)((size_t)(120 * 1024) / rec);               // tap records held in LDS together
    rmax = rmax < 1 ? 1 : (rmax > SCAT_RMAX ? SCAT_RMAX : rmax);
    const int smem = (int)(rmax * rec > stage ? rmax * rec : stage);
    const int build_smem = (int)(rec + (size_t)HW * 4 + 16);     // the record plus the per-pixel counters
    hipError_t e = hipSuccess;
    if (!a.taps_ready) {
        e = dg_set_max_smem(reinterpret_cast<const void*>(k_build_taps), build_smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_build_taps, dim3(a.B, 2), dim3(SCAT_THREADS), build_smem, s, a);
    }
    {
        // small sample grids: tiles of a 32-channel group + tap lists of several sources in LDS (k_scatter_small)
        const bool half = (a.DP / 32) * a.B * 2 < 512;               // 32-channel blocks would not even fill the CUs twice
        // ... and 16-channel blocks fewer than half the CUs (config 4's shard of 8 images: 96 blocks): 8 channels per block.  Measured:
        // config 4 shard 0.2106 -> 0.2071 ms; config 2 (192 blocks of 16 channels) 0.1939 -> 0.1950 and config 3 (512) 0.220 -> 0.231
        // the other way - the tap records are loaded once per block, whatever its channel count
        bool quarter = (a.DP / 16) * a.B * 2 < 128;
#ifdef DG_DEVTOOLS
        if (const char* sc = getenv("DG_SCAT_SMALL_CG")) quarter = atoi(sc) == 8;
#endif
        const int CGv = quarter ? 8 : (half ? 16 : 32);
        const size_t per_src = rec + (size_t)a.Ppad * (CGv + 1) * 4;
        int rb = (int)((size_t)(150 * 1024) / per_src);
        rb = rb > SCAT_RMAX ? SCAT_RMAX : rb;
        if (HW <= SCAT_THREADS && rb >= 2) {
            const int sm = (int)(rb * per_src);
            const dim3 sgrid(a.DP / CGv, a.B, 2);
            if (quarter) {
                e = dg_set_max_smem(reinterpret_cast<const void*>(k_scatter_small<8>), sm);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k_scatter_small<8>, sgrid, dim3(SCAT_THREADS), sm, s, a, rb);
            } else if (half) {
                e = dg_set_max_smem(reinterpret_cast<const void*>(k_scatter_small<16>), sm);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k_scatter_small<16>, sgrid, dim3(SCAT_THREADS), sm, s, a, rb);
            } else {
                e = dg_set_max_smem(reinterpret_cast<const void*>(k_scatter_small<32>), sm);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k_scatter_small<32>, sgrid, dim3(SCAT_THREADS), sm, s, a, rb);
            }
            return hipGetLastError();
        }
    }
    dim3 grid((a.D + SCAT_DC - 1) / SCAT_DC, a.B, 2);
    const int npass = (HW + SCAT_PX - 1) / SCAT_PX;
#define DG_SCAT(NP)                                                                                                      \
    {                                                                                                                    \
        e = dg_set_max_smem(reinterpret_cast<const void*>(k_scatter_grad<NP>), smem); \
        if (e != hipSuccess) return e;                                                                                   \
        hipLaunchKernelGGL(k_scatter_grad<NP>, grid, dim3(SCAT_THREADS), smem, s, a, rmax);                                    \
        return hipGetLastError();                                                                                        \
    }
    if (npass <= 2) DG_SCAT(2)
    if (npass <= 4) DG_SCAT(4)
    if (npass <= 8) DG_SCAT(8)
    DG_SCAT(16)
#undef DG_SCAT
}

// ------------------------------------------------------------------------------------------
// FPS.  One block per image.  All arithmetic is IEEE fp32 with the reference's operation order and
// no fused multiply-add, so that the selected set is bit-identical to numpy's for the same depth.
#define FPS_THREADS 512      // (8 waves: 2 points per thread at 28 x 28 - the packed fp32 distance update issues at ~8 cycles per
                             //  instruction, so halving the points per thread buys more than the wider barrier costs: 47 -> 42 us, round 5)
// (wave-wide reductions below: DPP row operations - inclusive scan inside each row of 16 lanes (row_shr 1,2,4,8), then the row
//  totals are carried across rows (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3); lane 63 holds the result)
// Pooled depth of output pixel (i, j) from image rows staged in LDS (`st` holds rows ybase.. of the depth map, W floats
// each): adaptive_avg_pool2d's window, summed row-major and sequentially, then divided by the window height and by the
// window width (the torch CPU operator's order, bit for bit).
__device__ __forceinline__ float fps_pool_lds(const float* st, int ybase, int H, int W, int h, int w, int i, int j) {
    const int ys = (i * H) / h, ye = ((i + 1) * H + h - 1) / h;
    const int xs = (j * W) / w, xe = ((j + 1) * W + w - 1) / w;
    float s = 0.f;
    for (int y = ys; y < ye; ++y) {
        const float* row = st + (size_t)(y - ybase) * W;
        int x = xs;
        for (; x + 8 <= xe; x += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = row[x + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = __fadd_rn(s, t[u]);
        }
        for (; x < xe; ++x) s = __fadd_rn(s, row[x]);
    }
    return __fdiv_rn(__fdiv_rn(s, (float)(ye - ys)), (float)(xe - xs));        // sum / kh / kw: the operator's two divisions
}

// Packed fp32 arithmetic of the distance update (two points per instruction; separately rounded multiply and add, exactly
// numpy's float32 operations: no fused multiply-add).  `l` carries two coordinates of the last selected point; LO / HI pick
// which of them is broadcast to both halves.
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
// wave-wide max of floats that are >= 0 or -1 (their bit patterns order like signed integers)
__device__ __forceinline__ int dpp_wave_max_i(int v) {
    const int small = (int)0x80000000;
#define DG_DPP_MAXI(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(small, v, ctrl, rmask, 0xf, false))
    DG_DPP_MAXI(0x111, 0xf); DG_DPP_MAXI(0x112, 0xf); DG_DPP_MAXI(0x114, 0xf); DG_DPP_MAXI(0x118, 0xf);
    DG_DPP_MAXI(0x142, 0xa); DG_DPP_MAXI(0x143, 0xc);
#undef DG_DPP_MAXI
    return __builtin_amdgcn_readlane(v, 63);
}

template <int NPT>       // points per thread (even): h*w <= NPT * FPS_THREADS
__global__ __launch_bounds__(FPS_THREADS) void k_fps_coords(const float* __restrict__ depth, const float* __restrict__ depth_b, int Ba,
                                                            int H, int W, int h, int w,
                                                            int S, float factor, int stage_floats, float* __restrict__ out_coords,
                                                            int32_t* __restrict__ out_inds, const float* __restrict__ pooled) {
    constexpr int NP2 = NPT / 2;
    static_assert(NPT % 2 == 0, "points are updated in pairs");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = h * w, nsel = S * S, nwords = (HW + 31) >> 5;
    int* order = reinterpret_cast<int*>(sm);               // [nsel] selection order
    uint32_t* selbits = reinterpret_cast<uint32_t*>(order + nsel);   // [nwords] selected set as a bit mask
    // the round's arg-max is formed by the LDS atomic unit: a signed 64-bit key (distance : 0x7fffffff - point index; the
    // larger key is the larger distance, then the lower index) per round, three keys in rotation
    __shared__ long long skey[3];
    __shared__ int tie_low;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    // (images Ba.. of the launch come from a second tensor: depth and depth_pos of one step, dg_fps_coords_pair)
    const float* d = n < Ba ? depth + (size_t)n * H * W : depth_b + (size_t)(n - Ba) * H * W;
    float4* const pts = reinterpret_cast<float4*>(selbits + nwords + ((4 - ((nsel + nwords) & 3)) & 3));   // [HW] {x, y, z, 0}: takes over the staging buffer

    // adaptive_avg_pool2d + depth2points; every thread keeps its points (idx = tid + FPS_THREADS*k) and their running
    // distances in registers; a point past the map has distance -1 and never wins.
    // The depth map goes through LDS in bands of whole pooled rows: coalesced, independent loads (one memory latency per
    // band instead of one per few addends of every window), then every thread sums the windows of its points from LDS.
    fps2 qx[NP2], qy[NP2], qz[NP2], qd[NP2];
#pragma unroll
    for (int k = 0; k < NPT; ++k) { qx[k >> 1][k & 1] = 0.f; qy[k >> 1][k & 1] = 0.f; qz[k >> 1][k & 1] = 0.f; qd[k >> 1][k & 1] = -1.f; }
    if (pooled) {
        // the pooled map comes from k_pool_depth (a launch over the whole chip in front of this one: one block per image streaming
        // its 200-KB depth map through one CU was 15 of this launch's 62 us): depth2points only
        const float* pd = pooled + (size_t)n * HW;
        float dvs[NPT];
#pragma unroll
        for (int k = 0; k < NPT; ++k) { const int idx = tid + FPS_THREADS * k; dvs[k] = idx < HW ? pd[idx] : 0.f; }
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int idx = tid + FPS_THREADS * k;
            const int i = idx / w, j = idx - i * w;
            if (idx < HW) {
                const float dv = dvs[k];
                const float fd = __fmul_rn(factor, dv);
                qy[k >> 1][k & 1] = __fdiv_rn(__fmul_rn(fd, __fsub_rn((float)i, (float)h / 2.0f)), (float)h);
                qx[k >> 1][k & 1] = __fdiv_rn(__fmul_rn(fd, __fsub_rn((float)j, (float)w / 2.0f)), (float)w);
                qz[k >> 1][k & 1] = __fmul_rn(-dv, 5.0f);
                qd[k >> 1][k & 1] = __builtin_inff();
            }
        }
    } else {
        float* stage = reinterpret_cast<float*>(selbits + nwords + ((4 - ((nsel + nwords) & 3)) & 3));     // 16-byte aligned
        const int max_rows = stage_floats / W;                      // image rows that fit the staging buffer
        const bool vec4 = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(d) & 15) == 0);
        for (int r0 = 0; r0 < h;) {
            // pooled rows r0 .. r1-1: image rows ys(r0) .. ye(r1-1)-1
            const int y0 = (r0 * H) / h;
            int r1 = r0 + 1;
            while (r1 < h && ((r1 + 1) * H + h - 1) / h - y0 <= max_rows) ++r1;
            const int y1 = (r1 * H + h - 1) / h;
            const float* src = d + (size_t)y0 * W;
            const int nfl = (y1 - y0) * W;
            if (vec4) {
                const int n4 = nfl >> 2;                             // (W % 4 == 0: whole rows are whole float4s)
                for (int q = tid; q < n4; q += FPS_THREADS * 8) {
                    f32x4 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = q + u * FPS_THREADS < n4 ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + q + u * FPS_THREADS) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (q + u * FPS_THREADS < n4) reinterpret_cast<f32x4*>(stage)[q + u * FPS_THREADS] = t[u];
                }
            } else {
                for (int q = tid; q < nfl; q += FPS_THREADS) stage[q] = src[q];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int idx = tid + FPS_THREADS * k;
                const int i = idx / w, j = idx - i * w;
                if (idx < HW && i >= r0 && i < r1) {
                    const float dv = fps_pool_lds(stage, y0, H, W, h, w, i, j);
                    const float fd = __fmul_rn(factor, dv);
                    qy[k >> 1][k & 1] = __fdiv_rn(__fmul_rn(fd, __fsub_rn((float)i, (float)h / 2.0f)), (float)h);
                    qx[k >> 1][k & 1] = __fdiv_rn(__fmul_rn(fd, __fsub_rn((float)j, (float)w / 2.0f)), (float)w);
                    qz[k >> 1][k & 1] = __fmul_rn(-dv, 5.0f);
                    qd[k >> 1][k & 1] = __builtin_inff();
                }
            }
            __syncthreads();
            r0 = r1;
        }
    }
    for (int k = tid; k < nwords; k += FPS_THREADS) selbits[k] = 0u;
    // (the last band's barrier is behind us: the staging buffer is free) the point table, then point 0 starts the selection
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int idx = tid + FPS_THREADS * k;
        if (idx < HW) pts[idx] = make_float4(qx[k >> 1][k & 1], qy[k >> 1][k & 1], qz[k >> 1][k & 1], 0.f);
    }
    if (tid < 3) skey[tid] = (long long)0x8000000000000000ull;
    if (lane == 0) order[0] = 0;
    __syncthreads();
    fps2 lxy = {pts[0].x, pts[0].y}, lzz = {pts[0].z, 0.f};
    const uint32_t key_addr = lds_addr(&skey[0]);
    int cur = 1, nxt = 2;                  // key of round it: it % 3
    // Rounds, one barrier each.  The loop is one dependent chain on one wave per SIMD: what counts is the instruction count, the
    // TAKEN branches (~40-80 cycles each) and the LDS round trips (scripts/micro/clock_light_load.hip, fps_round.hip), so: every
    // wave reduces the VALUES with DPP, the lanes that hold the wave's maximum (almost always one) hand {value, index} to an
    // LDS atomic max - which also breaks exact ties towards the lower index, as numpy's first-max over the ascending remainder
    // does - and after the barrier every thread reads the winning key and that point's coordinates.
    for (int it = 1; it < nsel; ++it) {
        float nd[NPT];
#pragma unroll
        for (int p = 0; p < NP2; ++p) {
            const fps2 dx = fps_sub_bcast<false>(lxy, qx[p]), dy = fps_sub_bcast<true>(lxy, qy[p]), dz = fps_sub_bcast<false>(lzz, qz[p]);
            const fps2 dd = fps_add(fps_add(fps_mul(dx, dx), fps_mul(dy, dy)), fps_mul(dz, dz));
            qd[p][0] = nd[2 * p] = fps_min(dd[0], qd[p][0]);
            qd[p][1] = nd[2 * p + 1] = fps_min(dd[1], qd[p][1]);
        }
        // this thread's largest distance (values are >= 0 or -1: their bit patterns order like signed integers)
        int mb = __float_as_int(nd[0]);
#pragma unroll
        for (int k = 1; k < NPT; ++k) mb = max(mb, __float_as_int(nd[k]));
        const int wmax = dpp_wave_max_i(mb);
        if (mb == wmax) {
            int bk = NPT - 1;                  // the FIRST of this thread's points that attains it (ascending k = ascending index)
#pragma unroll
            for (int k = NPT - 2; k >= 0; --k) bk = __float_as_int(nd[k]) == mb ? k : bk;
            const long long key = ((long long)mb << 32) | (long long)(0x7fffffff - (tid + FPS_THREADS * bk));
            // (asm: hipcc would wrap an atomicMax in a scalar loop over the active lanes - there is almost always exactly one)
            asm volatile("ds_max_i64 %0, %1" :: "v"(key_addr + 8 * cur), "v"(key) : "memory");
        }
        if (lane == 0) skey[nxt] = (long long)0x8000000000000000ull;     // (its readers of two rounds ago are behind the previous barrier)
        __syncthreads();
        const long long kb = skey[cur];
        int i0 = 0x7fffffff - (int)(kb & 0xffffffffll);
        if (__builtin_expect((int)(kb >> 32) == 0, 0)) {
            // Largest distance 0: every point that is left coincides with a selected one (e.g. a map of zero depth).  Selected
            // points are not marked in the registers - their distance simply became 0 in the round after their selection, which
            // never wins while any distance is positive - so here, and only here, they have to be told apart: the reference takes
            // the lowest index that is still unselected.
            if (tid == 0) tie_low = 0x7fffffff;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int idx = tid + FPS_THREADS * k;
                bool used = idx >= HW;
                for (int q = 0; q < it && !used; ++q) used = order[q] == idx;
                if (!used) atomicMin(&tie_low, idx);
            }
            __syncthreads();
            i0 = tie_low;
            __syncthreads();
        }
        const float4 bp = pts[i0];
        lxy[0] = bp.x; lxy[1] = bp.y; lzz[0] = bp.z;
        cur = nxt; nxt = nxt == 2 ? 0 : nxt + 1;
        if (lane == 0) order[it] = i0;         // (one lane of EVERY wave, the same value: the branch around it is never taken; kept in LDS - a global store per round would be waited for at every barrier)
    }
    __syncthreads();
    for (int k = tid; k < nsel; k += FPS_THREADS) {
        const int i0 = order[k];
        atomicOr(&selbits[i0 >> 5], 1u << (i0 & 31));
        if (out_inds) out_inds[(size_t)n * nsel + k] = i0;
    }
    __syncthreads();
    // selected set in row-major order -> coords (row/h, col/w)*2-1; rank of a pixel = selected pixels in front of it
    for (int idx = tid; idx < HW; idx += FPS_THREADS) {
        const uint32_t wd = selbits[idx >> 5];
        if (!((wd >> (idx & 31)) & 1u)) continue;
        int rank = __popc(wd & ((1u << (idx & 31)) - 1u));
        for (int k = 0; k < (idx >> 5); ++k) rank += __popc(selbits[k]);
        const int i = idx / w, j = idx - i * w;
        float* o = out_coords + ((size_t)n * nsel + rank) * 2;
        o[0] = __fsub_rn(__fmul_rn(__fdiv_rn((float)i, (float)h), 2.0f), 1.0f);
        o[1] = __fsub_rn(__fmul_rn(__fdiv_rn((float)j, (float)w), 2.0f), 1.0f);
    }
}

// adaptive_avg_pool2d of the depth maps to the feature map (src/modules.py:1003), one block per (pooled row, image): the image rows
// of the row's windows go through LDS (coalesced), one thread per output pixel sums its window in the operator's order
// (fps_pool_lds: the same function the sampler's in-kernel pooling uses - the pooled values have the same bits).
__global__ __launch_bounds__(256) void k_pool_depth(const float* __restrict__ depth, const float* __restrict__ depth_b, int Ba, int H, int W,
                                                    int h, int w, float* __restrict__ pooled) {
    extern __shared__ __attribute__((aligned(16))) float st[];
    const int i = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const float* d = n < Ba ? depth + (size_t)n * H * W : depth_b + (size_t)(n - Ba) * H * W;
    const int ys = (i * H) / h, ye = ((i + 1) * H + h - 1) / h, nfl = (ye - ys) * W;
    const float* src = d + (size_t)ys * W;
    if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0) {
        for (int q = tid; q < (nfl >> 2); q += 256) reinterpret_cast<f32x4*>(st)[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + q);
    } else {
        for (int q = tid; q < nfl; q += 256) st[q] = src[q];
    }
    __syncthreads();
    for (int j = tid; j < w; j += 256) pooled[((size_t)n * h + i) * w + j] = fps_pool_lds(st, ys, H, W, h, w, i, j);
}

hipError_t dg_launch_fps(const float* depth, const float* depth_b, int Ba, int B, int H, int W, int h, int w, int S, float factor,
                         float* out_coords, int32_t* out_inds, float* pooled_ws, hipStream_t s) {
    if (pooled_ws) {
        const int win_rows = (H + h - 1) / h + 1;
        if ((size_t)win_rows * W * 4 <= 64 * 1024) {
            hipLaunchKernelGGL(k_pool_depth, dim3(h, B), dim3(256), win_rows * W * 4, s, depth, depth_b, Ba, H, W, h, w, pooled_ws);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        } else {
            pooled_ws = nullptr;
        }
    }
    // LDS: selection order, selected-set mask, then the staging buffer for bands of the depth map: at least the image rows
    // of one pooled row, at most 128 KB
    const int head = (S * S + (h * w + 31) / 32 + 3) / 4 * 16;
    const int max_win_rows = (H + h - 1) / h + 1;
    if ((size_t)max_win_rows * W * 4 + head > 150 * 1024) return hipErrorInvalidValue;     // (a pooled row of a > 4k-wide map)
    const size_t whole = (size_t)H * W * 4;
    const int stage_bytes = (int)(whole < 128 * 1024 ? (whole + 15) / 16 * 16 : 128 * 1024);
    int stage_floats = (stage_bytes > max_win_rows * W * 4 ? stage_bytes : max_win_rows * W * 4) / 4;
    if (stage_floats < h * w * 4) stage_floats = h * w * 4;          // (the point table takes the buffer over)
    if ((size_t)stage_floats * 4 + head > 158 * 1024) return hipErrorInvalidValue;
    const int smem = head + stage_floats * 4;
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(B), dim3(FPS_THREADS), smem, s, depth, depth_b, Ba, H, W, h, w, S, factor, stage_floats, out_coords, out_inds,
                           (const float*)pooled_ws);
        return hipGetLastError();
    };
    if (h * w <= 2 * FPS_THREADS) return launch(k_fps_coords<2>);
    if (h * w <= 4 * FPS_THREADS) return launch(k_fps_coords<4>);
    if (h * w <= 8 * FPS_THREADS) return launch(k_fps_coords<8>);        // (4096 pixels: the sampler's limit, dg_api.hip fps_entry)
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// super_perm (src/modules.py:1184-1188) for `count` rows at once: rank of every key inside its row (ties by index) =
// position of that index in the argsort, then the fixed-point bump modulo B.  grid (count), block 256, LDS B floats.
// (Philox and the row body: dg_common.h dg_super_perm_row - the dense forward draws inside its first launch)
__global__ __launch_bounds__(256) void k_super_perms(const float* __restrict__ keys, uint64_t seed, unsigned long long* __restrict__ state,
                                                     int B, int64_t* __restrict__ out) {
    extern __shared__ float sk[];
    dg_super_perm_row(keys, seed, state, B, out, (int)blockIdx.x, (int)gridDim.x, sk);
}

hipError_t dg_launch_super_perms(const float* keys, uint64_t seed, unsigned long long* state, int count, int B, int64_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_super_perms, dim3(count), dim3(256), B * sizeof(float), s, keys, seed, state, B, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The random sample coordinates of a step (`torch.rand(B, S, S, 2) * 2 - 1`, twice: src/modules.py:1310-1321) from the
// device-resident generator of dg_super_perm_row: `n` floats in [-1, 1), keyed by {state[0], state[1]} READ FROM THE DEVICE; the
// block that finishes last advances the draw count, so the permutations drawn next (and the next step's coordinates) use another
// key.  For steps recorded in a hipGraph (cfg.dg_graph_safe): torch's own generator costs two launches per tensor there and two
// 64-bit fills per replay for its seed / offset.  Counters 2^31 .. : disjoint from the permutation keys' row * B + i.
// keep_p >= 0: Bernoulli(keep_p) keep flags (1 / 0) instead - the Dropout2d draws of a graph-recorded step (dg_rand_keep_state).
__global__ __launch_bounds__(256) void k_rand_coords_state(unsigned long long* __restrict__ state, float* __restrict__ out, int n, float keep_p) {
    const unsigned long long seed = state[0], draw = state[1];
    const uint64_t key = seed + 0x9E3779B97F4A7C15ull * draw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float u = (float)(dg_philox(key, 0x80000000u + (uint32_t)i) >> 8) * (1.0f / 16777216.0f);      // [0, 1)
        out[i] = keep_p >= 0.f ? (u < keep_p ? 1.f : 0.f) : u * 2.f - 1.f;
    }
    __syncthreads();                                        // (every thread of the block has read the state)
    if (threadIdx.x == 0) {
        // the block that finishes last advances the draw count (ticket in state[2], as dg_super_perm_row)
        __threadfence();
        if (atomicAdd(&state[2], 1ull) == (unsigned long long)gridDim.x - 1) {
            state[1] = draw + 1;
            state[2] = 0;
            __threadfence();
        }
    }
}
hipError_t dg_launch_rand_coords_state(unsigned long long* state, float* out, int n, hipStream_t s, float keep_p) {
    // (ten Philox rounds per value: one block was 15 us for the 74 k mask flags of a step; 256 blocks 8.2 us - every block ends with
    //  a ticket on ONE address, and 256 of those in a row are most of it; DG_RAND_BLOCKS in developer builds)
    int blocks = n >= 64 * 256 ? 64 : (n + 255) / 256;
#ifdef DG_DEVTOOLS
    if (const char* e = getenv("DG_RAND_BLOCKS")) blocks = atoi(e) > 0 ? atoi(e) : blocks;
#endif
    hipLaunchKernelGGL(k_rand_coords_state, dim3(blocks), dim3(256), 0, s, state, out, n, keep_p);
    return hipGetLastError();
}

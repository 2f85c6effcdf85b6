// The fused small-sample-grid path (round 5): Ppad <= 160 sampled positions per image - every recipe the reference ships runs here
// (feature_samples = 11 / 12, paper_reproduction.sh:5-14; src/modules.py:1287, 1304-1347) - at ANY feature width (FeaturePyramidNet's
// 2048 channels, src/modules.py:732-766).
//
// k_corr_small: ONE launch per call.  A block owns (image n, pair-set t[, half of the stationary tiles]) of helper()
// (src/modules.py:1231-1254) or the depth term (:1256-1278) and reads the SAMPLED fp32 rows of its two operands once:
//   phase 1  the feature rows (bf16 from the sampler) stream through LDS in chunks of 128 channels, their squared norms accumulated
//            beside them, fd_raw = <a_p, b_q> of the un-normalised rows on the bf16 MFMA, one accumulator tile per (R tile, S tile) held in
//            registers for the whole block; norm() (:789-790) is applied to the finished tiles as 1/|a_p| * 1/|b_q|;
//   phase 2  the code rows are normalised in fp32 and split into fp16 hi + lo parts; cd = hi.hi + hi.lo + lo.hi in one accumulator
//            (fp32-grade: the clamp mask 1[lo <= cd <= hi] is exact, for zero_clamp AND stabalize); pointwise centering from the
//            block's own row sums (:1236-1239: the row mean is local, old_mean is not - see below); epilogue in registers;
//            d/d(stationary code) += G^T y with the accumulator tile itself as the A operand; G goes to LDS (fp16) and comes back
//            row-wise as the A operand of d/d(streamed code) = G x, which gets its normalisation backward right here.
// old_mean (fd.mean() over the WHOLE batch, :1237) is not known inside a block.  Everything is linear in it:
//   loss sum  = sum clamp(cd) (fd' - shift) + old_mean * sum clamp(cd)               fd' = fd - rowmean
//   gradient  = [mask (fd' - shift)] . y     + old_mean * [mask] . y
// so a pointwise block emits both terms (the second set of gradient tiles costs MFMAs and bytes, no grid barrier); a one-wave
// launch behind the kernel (k_small_finish) reduces the partial sums, forms old_mean_t and the output scalars and leaves old_mean_t
// in `om`, which the backward tail multiplies into the weight of the second set (DgScatterSrc.dfac).
// Gradient tiles, partial sums and the operand-0 C part / inverse norms are written in the formats the existing backward tail
// (dg_post.hip k_grad_combine / k_scatter_small) reads.
//
// k_gather_rows: sample() (:822-825) of channel-last maps into the fp32 rows above, for calls whose maps k_plane_sample cannot take
// (code maps of another size than the feature maps, maps beyond its LDS planes).
#include "dg_common.h"
#include <hip/hip_runtime.h>

typedef int v4i_s __attribute__((ext_vector_type(4)));

#define SM_THREADS 256

// LDS images, all in 256-byte rows with granule (16 B) g of row r at slot g ^ (r & 15): the 16 lanes of one ds_read_b128 pass (16
// consecutive rows, one granule) cover all 64 banks.  Feature chunk: [row][128 bf16]; code: [row][<= 128 fp16]; -G: [S position][R
// position] fp16.
__device__ __forceinline__ uint32_t sm_c(int row, int g) { return (uint32_t)row * 256u + (uint32_t)((g ^ (row & 15)) << 4); }

// 1.0 where g != 0 (the clamp mask back out of the stored -G: the epilogue never stores an exact zero for an element that is on)
__device__ __forceinline__ f16x8 sm_mask_of(const f16x8 g) {
    const v4i_s b = __builtin_bit_cast(v4i_s, g) & v4i_s{0x7fff7fff, 0x7fff7fff, 0x7fff7fff, 0x7fff7fff};
    f16x8 t = __builtin_bit_cast(f16x8, b);
    const _Float16 big = (_Float16)32768.f;
    t = t * big;
    t = t * big;                                   // >= 1 (or inf) for every non-zero input, subnormals included
    f16x8 one;
#pragma unroll
    for (int e = 0; e < 8; ++e) one[e] = (_Float16)1.f;
    return __builtin_elementwise_min(t, one);
}

template <int NS, int NKD, bool PW>
__global__ __launch_bounds__(SM_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_corr_small(const DgSmallArgs a) {
    constexpr int NRM = NS == 5 ? 3 : NS;          // most stationary tiles of one block (5 tiles: blocks of 3 and 2)
    constexpr int NDF = NKD / 2;                   // 32-channel groups of a gradient tile row
    constexpr int KD = NKD * 16;
    constexpr int NJ = NRM + NS;                   // loader passes: 256 threads = one tile of 32 rows x 8 lanes
    constexpr int KC = 128;                        // channels per feature chunk: a 256-byte LDS row, two 16-byte pieces per thread and row
    // phase 1: two buffers of [R tiles | S tiles][32 rows][128 bf16]; phase 2: code images + the -G image (256-byte rows all)
    constexpr uint32_t FBUF = NJ * 32 * 256, FS_OFF = NRM * 32 * 256;
    constexpr uint32_t XH = 0, YH = NRM * 32 * 256, YL = YH + NS * 32 * 256, GB = YL + NS * 32 * 256; // (XL = GB until the epilogues)
    static_assert(2 * FBUF <= GB + NS * 32 * 256, "phase 1 fits the phase-2 image");
    extern __shared__ __attribute__((aligned(16))) char sm[];
    __shared__ __attribute__((aligned(16))) float invF[NJ * 32], invC[NJ * 32];
    __shared__ float red[4 * 4];

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = a.P, B = a.B, NT = a.Ppad >> 5;
    // block -> (image, pair-set, split): the blocks of one image sit on one XCD (ids are dealt round-robin over 8 XCDs), so the rows
    // of its stationary operand come out of one L2
    const int nsplit = a.nsplit;
    const int njob = a.mat ? 1 : a.T + (a.depth ? 1 : 0);
    const int per = njob * nsplit;
    const int xk = (int)blockIdx.x >> 3;
    const int n = ((int)blockIdx.x & 7) + 8 * (xk / per);
    if (n >= B) return;
    __shared__ unsigned long long span_keep[2];
    if (tid == 0) dg_span_enter(a.span, span_keep);
    const int jj = xk % per;
    const int sp = jj % nsplit;
    int t = jj / nsplit;
    if (a.mat) t = a.mat_t < 0 ? a.T : a.mat_t;
    const bool depth_job = t == a.T;
    const int rt0 = sp == 0 ? 0 : 3;
    const int NR = nsplit == 1 ? NS : (sp == 0 ? 3 : 2);
    const bool grad = a.grad != 0 && !a.mat;
    const int opS = depth_job ? 0 : a.opS[t];
    const int nS = (!depth_job && a.sidx[t]) ? (int)a.sidx[t][n] : n;       // image whose rows the streamed operand is (shared coordinates: the batch map)
#ifdef DG_DEVTOOLS
    const int abl = a.debug >> 4;          // developer timing ablations (results invalid): 1 no phase 2b, 2 no gradient work in 2a, 4 one feature chunk
#else
    constexpr int abl = 0;
#endif
#ifdef DG_DEVTOOLS
    unsigned long long stp[12];
    int nstp = 0;
#define SM_STAMP() do { if (a.debug == 1) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stp[nstp++] = wall_clock64(); } } while (0)
#else
#define SM_STAMP() do {} while (0)
#endif
    SM_STAMP();

    // ---- loader geometry: pass j = stationary tile j (j < NR) or streamed tile j - NR; thread = (row of the tile, lane of 8 channels)
    const int row32 = tid >> 3, gran = tid & 7;
    auto pass_pos = [&](int j) { return j < NR ? (rt0 + j) * 32 + row32 : (j - NR) * 32 + row32; };
    auto pass_row = [&](int j) { return (j < NR ? j : NRM + (j - NR)) * 32 + row32; };        // row of the LDS images, index into invF / invC

    f32x16 acc[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[st][i] = 0.f;

    // (the lambdas below capture copies, never `a` itself: a by-reference capture of the kernel argument makes hipcc copy the whole
    //  struct to scratch in every thread)
    const float* const rowsC_R = a.rowsC[0];
    const float* const rowsC_S = a.rowsC[opS];
    const int D4 = a.D4;

    // ---- the code rows: requested before the first feature chunk and normalised while that chunk is on its way (EARLY), kept as fp16
    //      hi + 2048 (x - hi) in registers until the LDS is free of feature chunks.  With 4 or 5 streamed tiles the registers do not
    //      take the code rows beside the staging of a feature chunk: there they are requested under the LAST chunk's MFMAs.
    constexpr bool EARLY = NS <= 3;
    f16x8 chi[2][NJ], clo[2][NJ];
    f32x4 vc[2][NJ][2];
    auto load_code = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool live = j < NR + NS;
            const int pos = live ? pass_pos(j) : 0;
            const float* src = (j < NR ? rowsC_R : rowsC_S) + ((size_t)(j < NR ? n : nS) * P + (pos < P ? pos : 0)) * D4;
            // (every load is issued unconditionally from a clamped address and zeroed afterwards: a load under a condition becomes
            //  a branch with its own s_waitcnt vmcnt(0) - the loads of a pass then run one memory latency after the other)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = c * 64 + gran * 8 + 4 * u;
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(src + (k < D4 ? k : D4 - 4));
                    const bool use = live && pos < P && k < D4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) vc[c][j][u][e] = use ? t4[e] : 0.f;
                }
        }
    };
    const bool write_x = grad && t == 0;           // the backward tail's copy of operand 0's normalised code (C part) and inverse norms
    char* const xop = a.xop;
    float* const xinv = a.xinv;
    const int blob_bytes = a.blob_bytes, blob_off_c = a.blob_off_c, Ppad = a.Ppad;
    auto norm_code = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j >= NR + NS) continue;
            const int pos = pass_pos(j);
            const bool okc = pos < P;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = fmaf(vc[c][j][u][e], vc[c][j][u][e], s);
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float inv = okc ? 1.f / fmaxf(sqrtf(s), DG_EPS_NORM) : 0.f;
            if (gran == 0) invC[pass_row(j)] = inv;
            const bool isR = j < NR;
#ifdef DG_DEVTOOLS
            if (a.debug == 2 && n == 0 && t == 1 && j == NR && row32 == 20)
                printf("row 20 gran %d: s %.9g inv %.9g v %.9g %.9g %.9g %.9g | %.9g %.9g %.9g %.9g\n", gran, s, inv, vc[0][j][0][0], vc[0][j][0][1],
                       vc[0][j][0][2], vc[0][j][0][3], vc[0][j][1][0], vc[0][j][1][1], vc[0][j][1][2], vc[0][j][1][3]);
#endif
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float xn[8];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xn[4 * u + e] = vc[c][j][u][e] * inv;
                        chi[c][j][4 * u + e] = (_Float16)xn[4 * u + e];
                    }
                // lo is formed from the hi BITS that are stored: hipcc otherwise converts the same value twice, once per use, with two
                // different instructions whose results differ at exact ties - hi from one rounding, lo from the other, 2.4e-4 off
                // (found against the reference fixture at C = 2048: one code row in 242 carried such an element)
                {
                    v4i_s hb = __builtin_bit_cast(v4i_s, chi[c][j]);
                    asm volatile("" : "+v"(hb));
                    chi[c][j] = __builtin_bit_cast(f16x8, hb);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) clo[c][j][e] = (_Float16)((xn[e] - (float)chi[c][j][e]) * 2048.f);
                const int g = c * 8 + gran;
                if (isR && write_x && g < KD / 8)
                    *reinterpret_cast<f16x8*>(xop + ((size_t)n * NT + rt0 + j) * blob_bytes + blob_off_c + (g * 32 + row32) * 16) = chi[c][j];
            }
            if (isR && write_x && gran == 0) xinv[(size_t)n * Ppad + pos] = inv;
        }
    };
    // (ONE call site per piece of straight-line code below - every copy of the loader or of the normalisation is a kilobyte of
    //  instructions, and the kernel has to stay inside the 64-KB instruction cache it shares with the neighbouring CU)
    const int C4 = a.C4, nch = depth_job ? 0 : ((abl & 4) ? 1 : (C4 + KC - 1) / KC);
    const __bf16* srcp[NJ];
    bool ok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const bool live = j < NR + NS;
        const int pos = live ? pass_pos(j) : 0;
        ok[j] = live && pos < P && !depth_job;
        srcp[j] = static_cast<const __bf16*>(a.rowsF[j < NR ? 0 : opS]) + ((size_t)(j < NR ? n : nS) * P + (ok[j] ? pos : 0)) * C4;
    }
    // The feature rows arrive as bf16 (the sampler's output): a chunk goes to LDS as it is, and the squared norms are taken from the
    // values that are multiplied.  Piece u of a thread = channels 64 u + 8 gran .. + 7 of its row.
    v4i_s v[NJ][2];
    float ss[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) ss[j] = 0.f;
    auto issue = [&](int ch) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // (no select at all: the rows are a whole number of chunks long - the sampler zero-fills the channels past C - and a
                //  padded position reads position 0's finite values, which its zero inverse norm and the epilogue's validity mask
                //  keep out of every result)
                v[j][u] = *reinterpret_cast<const v4i_s*>(srcp[j] + ch * KC + 64 * u + gran * 8);
            }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j >= NR + NS) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // both squares of a dword in one instruction.  (As asm: through __builtin_amdgcn_fdot2_f32_bf16 on a bit-cast
                    //  vector element hipcc 7.2 emitted all four instructions on element 0 - every norm wrong, 70 tests red)
                    const int w = v[j][u][e];
                    asm("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(ss[j]) : "v"(w));
                }
                *reinterpret_cast<v4i_s*>(sm + buf * FBUF + sm_c(pass_row(j), u * 8 + gran)) = v[j][u];
            }
        }
    };
    auto mfma_chunk = [&](int buf) {
        if (wid < NR) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(sm + buf * FBUF + sm_c(wid * 32 + r, 2 * ks + h));
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(sm + buf * FBUF + FS_OFF + sm_c(st * 32 + r, 2 * ks + h));
                    acc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[st], 0, 0, 0);
                }
            }
        }
    };
    SM_STAMP();
    if (!depth_job) issue(0);
    if (EARLY && !depth_job) load_code();          // (behind the first chunk's loads: one memory latency for both)
    // chunk ch + 1 (KC channels of every row) is in flight in registers while chunk ch is multiplied; buffer ch & 1 was last read two
    // iterations ago, in front of the previous iteration's barrier
#pragma unroll 1
    for (int ch = 0; ch + 1 < nch; ++ch) {
        stash(ch & 1);
        __syncthreads();
        issue(ch + 1);
        mfma_chunk(ch & 1);
    }
    // the last chunk, peeled: with 4-5 accumulator tiles the code rows' registers fit only here, where the staging registers die
    // (kept live around the loop they spill: 100-140 registers to scratch)
    if (nch > 0) { stash((nch - 1) & 1); __syncthreads(); }
    if (!EARLY || depth_job) load_code();
    if (nch > 0) mfma_chunk((nch - 1) & 1);
    SM_STAMP();
    norm_code();
    if (!depth_job) {
        // 1 / max(|row|, eps) of the feature rows (norm(), src/modules.py:789-790), from the fp32 values
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float s = ss[j];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            if (j < NR + NS && gran == 0) invF[pass_row(j)] = ok[j] ? 1.f / fmaxf(sqrtf(s), DG_EPS_NORM) : 0.f;
        }
    }
    __syncthreads();                               // every wave is done with the feature chunks
    SM_STAMP();

    // ---- the code images: [row][<= 128 fp16], hi and lo, stationary (X) and streamed (Y) rows
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j >= NR + NS) continue;
        const bool isR = j < NR;
        const int lrow = isR ? j * 32 + row32 : (j - NR) * 32 + row32;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int g = c * 8 + gran;
            if (g >= KD / 8) continue;
            *reinterpret_cast<f16x8*>(sm + (isR ? XH : YH) + sm_c(lrow, g)) = chi[c][j];
            *reinterpret_cast<f16x8*>(sm + (isR ? GB : YL) + sm_c(lrow, g)) = clo[c][j];
        }
    }
    __syncthreads();
    SM_STAMP();

    // ---- phase 2a: per stationary tile (wave) and streamed tile: cd, epilogue, -G to LDS, stationary-side gradient
    const bool active = wid < NR;
    f16x8 xl[NKD];
    if (active) {
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) xl[ks] = *reinterpret_cast<const f16x8*>(sm + GB + sm_c(wid * 32 + r, 2 * ks + h));
    }
    // selector fragments: B operand of the MFMA that brings 32 channels of 32 code rows into accumulator layout (rows in registers,
    // channel on the lane) - the layout of the gradient tiles, and, packed eight registers at a time, the B operand of the
    // gradient products (k <-> position in accumulator order): an exact transposition on the matrix core instead of sixty-four
    // 2-byte LDS gathers per tile
    f16x8 sel[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) sel[kk][j] = (16 * kk + 8 * h + j == r) ? (_Float16)1.f : (_Float16)0.f;
    auto tacc = [&](uint32_t base, int row0, int f) {
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const f16x8 af = *reinterpret_cast<const f16x8*>(sm + base + sm_c(row0 + r, 4 * f + 2 * kk + h));
            z = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, sel[kk], z, 0, 0, 0);
        }
        return z;
    };
    auto pack8 = [&](const f32x16& z, int q) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)z[8 * q + j];
        return o;
    };
    __syncthreads();                               // XL has been read by its wave: the region becomes the -G image

    f32x16 dR1[NDF], dR2[PW ? NDF : 1];
#pragma unroll
    for (int f = 0; f < NDF; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dR1[f][i] = 0.f; if (PW) dR2[f][i] = 0.f; }
    float L1 = 0.f, L2 = 0.f, sfd = 0.f, csum = 0.f;
    // (everything the tile loop reads from the kernel argument is copied to registers here: a scalar load inside the loop waits
    //  with lgkmcnt(0), i.e. for every LDS read in flight as well)
    const bool mat = a.mat != 0;
    float* const out_cd = a.out_cd;
    float* const out_loss = a.out_loss;
    const float lo = a.lo, hi = a.hi;
    const float shift = depth_job ? a.shift_depth : a.shift[t];
    const int pr = (rt0 + wid) * 32 + r;
    const bool rvalid = active && pr < P;
    if (depth_job) {
        // the depth term as a helper() whose fd is the rank-1 product of the depth indicators (dd = nz_p nz_q, src/modules.py:1273):
        // "fd_raw" = 1 with the indicators in the roles of the inverse norms - the tile loop below has no branch on the job kind
        for (int i = tid; i < NS * 32; i += SM_THREADS) invF[NRM * 32 + i] = a.nz[(size_t)n * a.Ppad + i];
#pragma unroll
        for (int st = 0; st < NS; ++st)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[st][i] = 1.f;
        __syncthreads();
    }
    const float iFr = !active ? 0.f : (depth_job ? (rvalid ? a.nz[(size_t)n * a.Ppad + pr] : 0.f) : invF[wid * 32 + r]);
    const float om_mat = (mat && PW && !depth_job) ? a.om[t] : 0.f;
    if (active) {
        float rm = 0.f;
        if (PW && !depth_job) {
            float rs = 0.f;
#pragma unroll
            for (int st = 0; st < NS; ++st)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const f32x4 iv = *reinterpret_cast<const f32x4*>(&invF[(NRM + st) * 32 + 8 * i4 + 4 * h]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rs = fmaf(acc[st][4 * i4 + e], iv[e], rs);
                }
            rs *= iFr;
            rs += __shfl_xor(rs, 32, 64);
            rm = rs / (float)P;
        }
        const float esub = rm + shift;               // e = fd - rowmean - shift
        if (PW && !depth_job) sfd = (h == 0 && rvalid) ? rm * (float)P : 0.f;      // sum fd of this lane's row (both halves hold it): old_mean's share
#pragma unroll 1
        for (int st = 0; st < NS; ++st) {
            // (a rolled loop - unrolled, the kernel outgrows the 64-KB instruction cache and is slower: 47.5 -> 50.9 us at config 3;
            //  the tile's accumulator is picked out of the register file with selects)
            f32x16 fdt = acc[0];
#pragma unroll
            for (int k = 1; k < NS; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) fdt[i] = st == k ? acc[k][i] : fdt[i];
            f32x16 cdv;
#pragma unroll
            for (int i = 0; i < 16; ++i) cdv[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKD; ++ks) {       // (channels D .. KD-1 are zero in every image: no guard, no branch)
                const f16x8 yh = *reinterpret_cast<const f16x8*>(sm + YH + sm_c(st * 32 + r, 2 * ks + h));
                const f16x8 yl = *reinterpret_cast<const f16x8*>(sm + YL + sm_c(st * 32 + r, 2 * ks + h));
                const f16x8 xh = *reinterpret_cast<const f16x8*>(sm + XH + sm_c(wid * 32 + r, 2 * ks + h));
                const f16x8 xs = xh * (_Float16)2048.f;                                    // exact (|x| <= 1)
                cdv = __builtin_amdgcn_mfma_f32_32x32x16_f16(yh, xs, cdv, 0, 0, 0);
                cdv = __builtin_amdgcn_mfma_f32_32x32x16_f16(yl, xh, cdv, 0, 0, 0);
                cdv = __builtin_amdgcn_mfma_f32_32x32x16_f16(yh, xl[ks], cdv, 0, 0, 0);
            }
            // the tile's 1 / |b_q| (depth term: nz_q) times this lane's 1 / |a_p|: element i of a lane = position (i & 3) + 8 (i >> 2) + 4 h
            float iFs[16];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 iv = *reinterpret_cast<const f32x4*>(&invF[(NRM + st) * 32 + 8 * i4 + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) iFs[4 * i4 + e] = iv[e] * iFr;
            }
            // Epilogue on pairs of elements.  Padded positions need no select in the sums: their code rows are zero, so cd = 0 and
            // clamp(cd) = 0 there (lo <= 0 <= hi); only the mask - which is ON at cd = 0 - has to know them.
            uint32_t gw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float e2[2];
                uint32_t m2 = 0u;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = 2 * j + u, s = st * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    const float fdn = fdt[i] * iFs[i];
                    const float e = fdn - esub;
                    const float cd = cdv[i] * (1.f / 2048.f);
                    const float cl = __builtin_amdgcn_fmed3f(cd, lo, hi);
                    L1 = fmaf(cl, e, L1);
                    L2 += cl;
                    csum += cd;
                    const bool on = rvalid && s < P && cd >= lo && cd <= hi;
                    e2[u] = e;
                    m2 |= on ? (u ? 0xffff0000u : 0x0000ffffu) : 0u;
                }
                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                h2_t hv;
                hv[0] = (_Float16)e2[0]; hv[1] = (_Float16)e2[1];
                uint32_t hw2 = __builtin_bit_cast(uint32_t, hv);
                if (PW) {        // an element that is on never stores an exact zero (the mask is read back out of -G): |g| = max(|g|, 2^-24)
                    const uint32_t ab = hw2 & 0x7fff7fffu;
                    typedef unsigned short u2_t __attribute__((ext_vector_type(2)));
                    const u2_t mx = __builtin_elementwise_max(__builtin_bit_cast(u2_t, ab), u2_t{1, 1});
                    hw2 = (hw2 & 0x80008000u) | __builtin_bit_cast(uint32_t, mx);
                }
                gw[j] = hw2 & m2;
            }
            uint16_t gh[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) { gh[2 * j] = (uint16_t)(gw[j] & 0xffffu); gh[2 * j + 1] = (uint16_t)(gw[j] >> 16); }
            if (mat) {       // dg_corr_materialize: the un-reduced tensors (ONE uniform branch per tile: the values are formed again)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int s = st * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (rvalid && s < P) {
                        const float fdn = fdt[i] * iFs[i], cd = cdv[i] * (1.f / 2048.f);
                        const size_t o = ((size_t)n * P + pr) * P + s;
                        if (out_cd) out_cd[o] = depth_job ? fdn : cd;
                        if (out_loss) out_loss[o] = -fminf(fmaxf(cd, lo), hi) * (fdn - esub + om_mat);
                    }
                }
            }
            if (grad && !(abl & 2)) {
                f16x8 ga[2], gm[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) ga[i >> 3][i & 7] = __builtin_bit_cast(_Float16, gh[i]);
                if (PW) { gm[0] = sm_mask_of(ga[0]); gm[1] = sm_mask_of(ga[1]); }
                if (!depth_job) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int s = st * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        *reinterpret_cast<uint16_t*>(sm + GB + sm_c(s, wid * 4 + (r >> 3)) + (r & 7) * 2) = gh[i];
                    }
                }
                // dR[r][:] += sum_s G[s][r] y[s][:]: the accumulator tile is the A operand (k <-> s in accumulator order), the B operand
                // is the streamed tile's code transposed on the matrix core
#pragma unroll
                for (int f = 0; f < NDF; ++f) {
                    const f32x16 yt = tacc(YH, st * 32, f);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f16x8 b = pack8(yt, q);
                        dR1[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[q], b, dR1[f], 0, 0, 0);
                        if (PW) dR2[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gm[q], b, dR2[f], 0, 0, 0);
                    }
                }
            }
        }
        if (grad) {
            // raw gradient tiles, accumulator order (dg_gtile_off): 1 KiB per wave instruction
            float* o1 = a.dRA[t] + (((size_t)n * NT + rt0 + wid) * NDF) * 1024 + lane * 4;
            float* o2 = (PW && !depth_job) ? a.dRA2[t] + (((size_t)n * NT + rt0 + wid) * NDF) * 1024 + lane * 4 : nullptr;
#pragma unroll
            for (int f = 0; f < NDF; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<f32x4*>(o1 + f * 1024 + g * 256) = f32x4{dR1[f][4 * g], dR1[f][4 * g + 1], dR1[f][4 * g + 2], dR1[f][4 * g + 3]};
                    if (PW && o2) *reinterpret_cast<f32x4*>(o2 + f * 1024 + g * 256) = f32x4{dR2[f][4 * g], dR2[f][4 * g + 1], dR2[f][4 * g + 2], dR2[f][4 * g + 3]};
                }
        }
    }
    SM_STAMP();
    // block partial sums (fixed order: butterfly inside the wave, waves 0..3 in turn)
    {
        float v4[4] = {L1, L2, sfd, csum};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v4[k] += __shfl_xor(v4[k], o, 64);
            if (lane == 0) red[wid * 4 + k] = v4[k];
        }
    }
    __syncthreads();                               // (also: the -G image is complete)
    if (a.mat) return;
    if (tid == 0) {                                // (one thread: its fence below orders these stores before its ticket)
#pragma unroll
        for (int k = 0; k < 4; ++k) a.part[(((size_t)t * B + n) * nsplit + sp) * 4 + k] = red[k] + red[4 + k] + red[8 + k] + red[12 + k];
    }

    // ---- phase 2b: d/d(streamed code) = G x over this block's stationary tiles, normalisation backward, final tiles
    if (grad && !depth_job && !(abl & 1)) {
        for (int st = wid; st < NS; st += 4) {
            f32x16 dS1[NDF], dS2[PW ? NDF : 1];
#pragma unroll
            for (int f = 0; f < NDF; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) { dS1[f][i] = 0.f; if (PW) dS2[f][i] = 0.f; }
#pragma unroll 1
            for (int rt = 0; rt < NR; ++rt) {
                // A = row s of the -G image, k <-> stationary position in accumulator order: two 8-byte runs per k-step
                f16x8 af[2], am[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f16x4 a0 = *reinterpret_cast<const f16x4*>(sm + GB + sm_c(st * 32 + r, rt * 4 + 2 * q) + 8 * h);
                    const f16x4 a1 = *reinterpret_cast<const f16x4*>(sm + GB + sm_c(st * 32 + r, rt * 4 + 2 * q + 1) + 8 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { af[q][e] = a0[e]; af[q][4 + e] = a1[e]; }
                    if (PW) am[q] = sm_mask_of(af[q]);
                }
#pragma unroll
                for (int f = 0; f < NDF; ++f) {
                    const f32x16 xt = tacc(XH, rt * 32, f);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f16x8 b = pack8(xt, q);
                        dS1[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[q], b, dS1[f], 0, 0, 0);
                        if (PW) dS2[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[q], b, dS2[f], 0, 0, 0);
                    }
                }
            }
            // norm() backward with the streamed rows y (hi + lo, brought into accumulator layout like the products): d = inv (g - y <g, y>);
            // element i of a lane = position (i&3) + 8 (i>>2) + 4 h of the tile, channel 32 f + r: the dot product runs over the 32
            // lanes of the half wave
            f32x16 yv[NDF];
#pragma unroll
            for (int f = 0; f < NDF; ++f) {
                const f32x16 yh = tacc(YH, st * 32, f), yl = tacc(YL, st * 32, f);
#pragma unroll
                for (int i = 0; i < 16; ++i) yv[f][i] = fmaf(yl[i], 1.f / 2048.f, yh[i]);
            }
            float* o1 = a.dRB[t][sp] + (((size_t)n * NT + st) * NDF) * 1024 + lane * 4;
            float* o2 = PW ? a.dRB2[t][sp] + (((size_t)n * NT + st) * NDF) * 1024 + lane * 4 : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 iv4 = *reinterpret_cast<const f32x4*>(&invC[(NRM + st) * 32 + 8 * g + 4 * h]);
                f32x4 w1[NDF], w2[PW ? NDF : 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g + e;
                    float d1 = 0.f, d2 = 0.f;
#pragma unroll
                    for (int f = 0; f < NDF; ++f) {
                        d1 = fmaf(dS1[f][i], yv[f][i], d1);
                        if (PW) d2 = fmaf(dS2[f][i], yv[f][i], d2);
                    }
                    d1 = half_sum(d1);
                    if (PW) d2 = half_sum(d2);
#pragma unroll
                    for (int f = 0; f < NDF; ++f) {
                        w1[f][e] = iv4[e] * (dS1[f][i] - yv[f][i] * d1);
                        if (PW) w2[f][e] = iv4[e] * (dS2[f][i] - yv[f][i] * d2);
                    }
                }
#pragma unroll
                for (int f = 0; f < NDF; ++f) {
                    *reinterpret_cast<f32x4*>(o1 + f * 1024 + g * 256) = w1[f];
                    if (PW) *reinterpret_cast<f32x4*>(o2 + f * 1024 + g * 256) = w2[f];
                }
            }
        }
    }

    if (tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dg_span_exit(a.span, span_keep); }
    SM_STAMP();
#ifdef DG_DEVTOOLS
    if (a.debug == 1 && blockIdx.x == 0 && tid == 0) {
        // wall_clock64 ticks at 100 MHz: 10 ns per tick
        printf("k_corr_small block 0 (NS %d NKD %d PW %d): setup %llu, chunk loop %llu, norm %llu, code images %llu, phase 2a %llu, phase 2b %llu  [x10 ns]\n",
               NS, NKD, (int)PW, stp[1] - stp[0], stp[2] - stp[1], stp[3] - stp[2], stp[4] - stp[3], stp[5] - stp[4], stp[6] - stp[5]);
    }
#endif
}

// The partial sums of the call -> the output scalars, old_mean_t for the backward.  Its own one-wave launch: a last-block ticket inside
// k_corr_small needs a device-scope release fence per block (the blocks sit on eight XCDs whose L2s are not coherent with each
// other), and every one of those fences writes the XCD's whole dirty L2 back - the gradient tiles the kernel has just stored:
// 35 of the launch's 51 us at config 2.
__global__ __launch_bounds__(64) void k_small_finish(const DgSmallArgs a) {
    // eight lanes per job (pair-set or depth term): lane (job = lane >> 3, sub = lane & 7) sums the partials sub, sub + 8, ... in double
    // (fixed order), three butterfly steps finish the job; jobs 8 .. 10 (more than six negatives) take a second round
    __shared__ double jobsum[DG_MAX_NEG + 3][4];
    const int lane = threadIdx.x, B = a.B, P = a.P;
    const int njob = a.T + (a.depth ? 1 : 0), nb = B * a.nsplit;
    for (int j0 = 0; j0 < njob; j0 += 8) {
        const int tt = j0 + (lane >> 3), sub = lane & 7;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        if (tt < njob)
            for (int i = sub; i < nb; i += 8) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(a.part + ((size_t)tt * nb + i) * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) s4[k] += (double)q4[k];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s4[k] += __shfl_xor(s4[k], 1, 64); s4[k] += __shfl_xor(s4[k], 2, 64); s4[k] += __shfl_xor(s4[k], 4, 64);
            if (sub == 0 && tt < njob) jobsum[tt][k] = s4[k];
        }
    }
    double m = 0.0;
    if (a.depth && a.nzsum) {                  // mean(dd) = mean_n (sum_p nz[n][p])^2 / P^2
        for (int i = lane; i < B; i += 64) { const double s = a.nzsum[i]; m += s * s; }
        for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
    }
    __syncthreads();
    if (lane == 0) {
        double accv[DG_OUT_COUNT];
#pragma unroll
        for (int i = 0; i < DG_OUT_COUNT; ++i) accv[i] = 0.0;
        const double numel = (double)B * P * P;
        for (int tt = 0; tt < njob; ++tt) {
            const double* s4 = jobsum[tt];
            if (tt < a.T) {
                const double om = a.pointwise ? s4[2] / numel : 0.0;
                const double l = s4[0] + om * s4[1];
                const int slot = tt < 2 ? tt : DG_OUT_LOSS_NEG;
                const double scale = tt < 2 ? 1.0 / numel : 1.0 / (numel * (a.T - 2));
                accv[slot] += -l * scale;
                accv[DG_OUT_CD_INTRA + (tt < 2 ? tt : 2)] += s4[3] * scale;
                a.om[tt] = (float)om;
            } else {
                accv[DG_OUT_LOSS_DEPTH] += -s4[0] / numel;
            }
        }
        if (a.depth && a.nzsum) accv[DG_OUT_DD] = m / numel;
        accv[DG_OUT_TOTAL] = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) accv[DG_OUT_TOTAL] += (double)a.wtot[i] * (double)(float)accv[i];
#pragma unroll
        for (int i = 0; i < DG_OUT_COUNT; ++i) a.out[i] = (float)accv[i];
    }
}

// LDS bytes of the phase-2 image (the larger one)
static int small_smem(int NS) {
    const int NRM = NS == 5 ? 3 : NS;
    return (NRM + 3 * NS) * 32 * 256;
}

bool dg_small_supported(int Ppad, int KD) { return Ppad >= 32 && Ppad <= 160 && (KD == 96 || KD == 128); }

hipError_t dg_launch_corr_small(const DgSmallArgs& a, hipStream_t s) {
    const int NS = a.Ppad / 32;
    if (!dg_small_supported(a.Ppad, a.KD) || a.nsplit != (NS == 5 ? 2 : 1)) return hipErrorInvalidValue;
    const int njob = a.mat ? 1 : a.T + (a.depth ? 1 : 0);
    const int per = njob * a.nsplit;
    const int grid = ((a.B + 7) / 8) * 8 * per;
    const int smem = small_smem(NS);
    auto go = [&](auto kern) -> hipError_t {
        const hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SM_THREADS), smem, s, a);
        return hipGetLastError();
    };
#define DG_SM_CASE(NS_)                                                                                  \
    if (NS == NS_) {                                                                                     \
        if (a.KD == 96) return a.pointwise ? go(k_corr_small<NS_, 6, true>) : go(k_corr_small<NS_, 6, false>); \
        return a.pointwise ? go(k_corr_small<NS_, 8, true>) : go(k_corr_small<NS_, 8, false>);          \
    }
    DG_SM_CASE(1) DG_SM_CASE(2) DG_SM_CASE(3) DG_SM_CASE(4) DG_SM_CASE(5)
#undef DG_SM_CASE
    return hipErrorInvalidValue;
}

hipError_t dg_launch_small_finish(const DgSmallArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_small_finish, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// sample() (src/modules.py:822-825: grid_sample, bilinear, padding_mode='border', align_corners=True, coordinates transposed) of
// channel-last fp32 maps into rows [image][position][K4].  One block = 8 positions of one (job, image); 32 lanes per position walk
// the channels four at a time.  Tap arithmetic as in k_plane_sample / k_gather_norm (the same operations in the same order).
__global__ __launch_bounds__(256) void k_gather_rows(const DgGatherRowsArgs a) {
    const int job = blockIdx.z, n = blockIdx.y, p = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (p >= a.P) return;
    const int K4 = a.K4[job], hh = a.h[job], ww = a.w[job];
    const int ii = p / a.S, j = p - ii * a.S;
    const float* cc = a.coords[job] + (((size_t)n * a.S + j) * a.Sh + ii) * 2;
    float x = ((cc[0] + 1.f) / 2.f) * (float)(ww - 1);
    float y = ((cc[1] + 1.f) / 2.f) * (float)(hh - 1);
    x = fminf(fmaxf(x, 0.f), (float)(ww - 1));
    y = fminf(fmaxf(y, 0.f), (float)(hh - 1));
    const float x0f = floorf(x), y0f = floorf(y);
    const float wx1 = x - x0f, wy1 = y - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const bool inx = x0 + 1 <= ww - 1, iny = y0 + 1 <= hh - 1;
    const float w00 = wy0 * wx0, w01 = inx ? wy0 * wx1 : 0.f, w10 = iny ? wy1 * wx0 : 0.f, w11 = (inx && iny) ? wy1 * wx1 : 0.f;
    const int pix = y0 * ww + x0, dx = inx ? 1 : 0, dy = iny ? ww : 0;
    const int img = a.srcidx[job] ? (int)a.srcidx[job][n] : n;
    const float* base = a.src[job] + (size_t)img * hh * ww * K4;
    const float* t00 = base + (size_t)pix * K4;
    const float* t01 = base + (size_t)(pix + dx) * K4;
    const float* t10 = base + (size_t)(pix + dy) * K4;
    const float* t11 = base + (size_t)(pix + dy + dx) * K4;
    const int Kout = a.Kout[job];
    const bool as16 = a.as_bf16[job] != 0;
    float* out = static_cast<float*>(a.rows[job]) + ((size_t)n * a.P + p) * Kout;
    __bf16* out16 = static_cast<__bf16*>(a.rows[job]) + ((size_t)n * a.P + p) * Kout;
    for (int k = 4 * l; k < Kout; k += 128) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (k < K4) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(t00 + k), a1 = *reinterpret_cast<const f32x4*>(t01 + k);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(t10 + k), a3 = *reinterpret_cast<const f32x4*>(t11 + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = a0[e] * w00;
                acc = fmaf(a1[e], w01, acc);
                acc = fmaf(a2[e], w10, acc);
                acc = fmaf(a3[e], w11, acc);
                o[e] = acc;
            }
        }
        if (as16) {
            bf16x4 ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = (__bf16)o[e];
            *reinterpret_cast<bf16x4*>(out16 + k) = ob;
        } else {
            *reinterpret_cast<f32x4*>(out + k) = o;
        }
    }
}

hipError_t dg_launch_gather_rows(const DgGatherRowsArgs& a, hipStream_t s) {
    if (a.njobs < 1) return hipSuccess;
    hipLaunchKernelGGL(k_gather_rows, dim3((a.P + 7) / 8, a.B, a.njobs), dim3(256), 0, s, a);
    return hipGetLastError();
}

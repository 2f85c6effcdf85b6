// Fused correlation-loss kernel, one-wave-per-SIMD form (gfx950 / MI355X): helper() of the reference
// (src/modules.py:1231-1254) for the pair-sets whose stationary operand is operand 1, gradient pass of the
// zero_clamp / no-stabalize recipe.
//
// One workgroup = 4 waves = one per SIMD, each with the whole 512-entry register file: a wave owns TWO 32-row tiles of the
// stationary operand R (64 rows).  Their normalised bf16 feature fragments (2 x NKF x 4 registers) live in the ACCUMULATOR
// half of the register file for the whole block and are named literally as the B operands of the feature MFMAs (inline
// asm: hipcc has no way to be told "these 192 registers are MFMA operands and never move"; left to itself it parks them
// there and copies every one back per use).  Four of the six gradient accumulators sit there too.  The arch VGPRs hold
// the two fd / cd accumulator pairs, the other two gradient accumulators, the LDS fragment ring and the epilogue.
//
// Per streamed tile of 32 positions of S (DMA'd into one of three LDS buffers two tiles ahead, as in dg_corr.hip) a wave
// issues 70 MFMAs in four phases, and every non-MFMA instruction of the tile is placed in one of their issue gaps:
//     A  chain of fragment 0: 24 x bf16 (fd) + 5 x f16 (cd)        gaps: the 9 LDS-DMA pieces of tile t+2, fd init of fragment 1
//     B  chain of fragment 1                                       gaps: epilogue of fragment 0 (mask, -G, fp16 pack, G store),
//                                                                        the six B fragments of the gradient product
//        -- counted vmcnt + the one workgroup barrier of the tile: tile t+1 visible, buffer of tile t free --
//     C  dR_0 += G_0^T ScP (6 MFMAs, accumulator tile as A operand)  gaps: first fragments of tile t+1, epilogue of fragment 1
//     D  dR_1 += G_1^T ScP (6 MFMAs)                                 gaps: rest of that epilogue, G store, fd init of fragment 0
// An accumulator is read by the VALU no sooner than two MFMAs after the chain that wrote it (the asm MFMAs are invisible to
// hipcc's hazard recogniser).  Outputs are those of k_corr_main: G tiles (fp16) for k_gs, raw gradient tiles, block sums.
//
// Round 4: the workgroups are PERSISTENT (one per CU, walking the work items).  BUILD THIS FILE WITH `-mllvm -disable-machine-licm`
// (the Makefile's rule for dg_corr2.o, scripts/build_variant.sh, the audit in tests/test_host_cpu.py): without it hipcc hoists
// constants of the item body out of the walk and spills scalar registers into VGPR lanes (correct, slower).
#include "dg_common.h"
#include <utility>
#include <cstdio>
#include <cstdlib>

typedef int v4i_t __attribute__((ext_vector_type(4)));
#define C2_RED_BYTES 256   // behind the tile buffers: the block-end reduction's [8 fragment slots][4] dwords + [8] output pointers
#ifndef C2_PF
#define C2_PF 8            // LDS fragment reads in flight ahead of the MFMA that consumes them (8, or 12 in developer builds of the exact-mask form)
#endif

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- inline-asm building blocks (register numbers are literal: the accumulator file is owned by this kernel)
template <int I> __device__ __forceinline__ void agpr_load16(const void* p) {      // a[4I..4I+3] <- 16 bytes at p
    asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off" :: "v"(p), "n"(4 * I), "n"(4 * I + 3) : "memory");
}
template <int I> __device__ __forceinline__ void mfma_fd(f32x16& acc, const v4i_t& a) {     // acc += A x Rf (bf16, B = a[4I..4I+3])
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "n"(4 * I), "n"(4 * I + 3));
}
__device__ __forceinline__ void mfma_h(f32x16& acc, const v4i_t& a, const v4i_t& b) {         // acc += A x B (f16)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_h0(f32x16& acc, const v4i_t& a, const v4i_t& b) {        // acc = A x B (f16)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
template <int X> __device__ __forceinline__ void mfma_h_acc(const v4i_t& a, const v4i_t& b) { // a[X..X+15] += A x B (f16); A must not be fresh from the VALU (2 wait states)
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "n"(X), "n"(X + 15));
}
template <int X> __device__ __forceinline__ float agpr_read(void) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "n"(X));
    return v;
}
template <int X> __device__ __forceinline__ void agpr_zero4(void) {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0"
                 :: "n"(X), "n"(X + 1), "n"(X + 2), "n"(X + 3));
}
template <int X> __device__ __forceinline__ void agpr_store16_nt(void* p) {                   // 16 bytes a[X..X+3] -> p (non-temporal)
    asm volatile("global_store_dwordx4 %0, a[%c1:%c2], off nt" :: "v"(p), "n"(X), "n"(X + 3) : "memory");
}
__device__ __forceinline__ void wait_vm_lgkm_barrier(int n) {     // tile landed (this wave's pieces), own LDS reads done, workgroup barrier
    switch (n) {
#define DG_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        DG_W(0) DG_W(1) DG_W(2) DG_W(3) DG_W(4) DG_W(5) DG_W(6) DG_W(7) DG_W(8) DG_W(9) DG_W(10) DG_W(11) DG_W(12) DG_W(13)
        DG_W(14) DG_W(15) DG_W(16) DG_W(17) DG_W(18) DG_W(19) DG_W(20) DG_W(21) DG_W(22) DG_W(23) DG_W(24) DG_W(25) DG_W(26) DG_W(27)
        DG_W(28) DG_W(29) DG_W(30) DG_W(31) DG_W(32) DG_W(33) DG_W(34) DG_W(35) DG_W(36) DG_W(37) DG_W(38) DG_W(39) DG_W(40)
#undef DG_W
        default: asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
}
// Epilogue of four accumulator elements: -G = cd >= 0 ? fd'' - shift : 0, packed to fp16 (round to nearest even).  Four
// compares into four scalar mask pairs first, then the selects: no select waits on the compare just ahead of it (hipcc's own
// sequence is compare - s_nop 1 - select per element, 7 VALU per pair).  10 VALU per four elements.
__device__ __forceinline__ void epi4(const float y0, const float y1, const float y2, const float y3, const float c0, const float c1,
                                     const float c2, const float c3, int& o0, int& o1) {
    unsigned long long m0, m1, m2, m3;
    int t;
    asm volatile("v_cmp_le_f32_e64 %3, 0, %11\n\t"
                 "v_cmp_le_f32_e64 %4, 0, %12\n\t"
                 "v_cmp_le_f32_e64 %5, 0, %13\n\t"
                 "v_cmp_le_f32_e64 %6, 0, %14\n\t"
                 "v_cndmask_b32_e64 %0, 0, %7, %3\n\t"
                 "v_cndmask_b32_e64 %2, 0, %8, %4\n\t"
                 "v_cndmask_b32_e64 %1, 0, %9, %5\n\t"
                 "v_cvt_pk_f16_f32 %0, %0, %2\n\t"
                 "v_cndmask_b32_e64 %2, 0, %10, %6\n\t"
                 "v_cvt_pk_f16_f32 %1, %1, %2"
                 : "=&v"(o0), "=&v"(o1), "=&v"(t), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
                 : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
}

// every accumulator register is named somewhere in this kernel: tell hipcc (kernel descriptor, no compiler use)
#define DG_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
__device__ __forceinline__ void declare_agprs(void) {
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", DG_A8(1), DG_A8(2), DG_A8(3), DG_A8(4), DG_A8(5),
                 DG_A8(6), DG_A8(7), DG_A8(8), DG_A8(9), DG_A8(10), DG_A8(11), DG_A8(12), DG_A8(13), DG_A8(14), DG_A8(15), DG_A8(16),
                 DG_A8(17), DG_A8(18), DG_A8(19), DG_A8(20), DG_A8(21), DG_A8(22), DG_A8(23), DG_A8(24), "a250", "a251", "a252",
                 "a253", "a254", "a255");
}

// ---- hand-placed loop instructions (nothing in the tile loop is left to hipcc's own wait insertion)
typedef double acc_t __attribute__((ext_vector_type(8)));       // a 32x32 accumulator as eight 64-bit halves (v_mov_b64 initialisation)
template <int OFF> __device__ __forceinline__ void lds_rd(v4i_t& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void wait_lgkm(void) { asm volatile("s_waitcnt lgkmcnt(%c0)" :: "n"(N) : "memory"); }
#ifdef C2_XNOP      // (round-4 experiment: what one more s_nop 0 per chain MFMA costs = what the ones hipcc inserts cost)
#define C2_XN "s_nop 0\n\t"
#else
#define C2_XN
#endif
template <int I> __device__ __forceinline__ void mfma_fd8(acc_t& acc, const v4i_t& a) {
    asm volatile(C2_XN "v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "n"(4 * I), "n"(4 * I + 3));
}
// first MFMA of an fd chain: the accumulator STARTS at c (sixteen copies of the lane's c0, kept in registers for the whole block) -
// no per-tile re-initialisation of the accumulator by sixteen 64-bit moves (round 4: hipcc gathered them into bursts of eight in
// one MFMA gap)
template <int I> __device__ __forceinline__ void mfma_fd8_from(acc_t& acc, const v4i_t& a, const acc_t& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(acc) : "v"(a), "v"(c), "n"(4 * I), "n"(4 * I + 3));
}
__device__ __forceinline__ void mfma_h8(acc_t& acc, const v4i_t& a, const v4i_t& b) {
    asm volatile(C2_XN "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_h80(acc_t& acc, const v4i_t& a, const v4i_t& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
// one 1-KiB LDS-DMA piece: M0 = LDS destination (written in this statement), the vector add is the wait state between the M0
// write and the DMA; source = scalar tile base + per-lane offset
template <int KOFF> __device__ __forceinline__ void dma_piece(uint32_t lds_dst, uint32_t voff, const char* sbase) {
    uint32_t tmp;
    asm volatile("s_add_u32 m0, %1, %c3\n\tv_add_u32 %0, %c3, %2\n\tglobal_load_lds_dwordx4 %0, %4"
                 : "=&v"(tmp) : "s"(lds_dst), "v"(voff), "n"(KOFF), "s"(sbase) : "memory", "scc");     // s_add_u32 writes SCC
}
// the same piece in two halves for two neighbouring MFMA gaps (M0 stays as set: nothing between the two touches it)
template <int KOFF> __device__ __forceinline__ void dma_setup(uint32_t lds_dst, uint32_t voff, uint32_t& tmp) {
    asm volatile("s_add_u32 m0, %1, %c3\n\tv_add_u32 %0, %c3, %2" : "=v"(tmp) : "s"(lds_dst), "v"(voff), "n"(KOFF) : "memory", "scc");
}
__device__ __forceinline__ void dma_go(uint32_t tmp, const char* sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(tmp), "s"(sbase) : "memory");
}
// Epilogue of two accumulator elements: -G = cd >= 0 ? fd'' - shift : 0 as one packed fp16 word, 4 VALU, no scalar masks:
// the two sign halves of cd are gathered (v_perm), smeared over their 16-bit halves (packed arithmetic shift) and clear the
// packed fp16 pair (v_bfi).  cd is never -0 (the cd chain starts at +0 and adds products: x + (-x) = +0), so "sign bit clear"
// is the reference's `cd >= 0`.
__device__ __forceinline__ int epi2(const float y0, const float y1, const float c0, const float c1, const uint32_t sel) {
    int o, t;
    asm volatile("v_perm_b32 %1, %5, %4, %6\n\t"
                 "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                 "v_pk_ashrrev_i16 %1, 15, %1 op_sel_hi:[0,1]\n\t"
                 "v_bfi_b32 %0, %1, 0, %0"
                 : "=&v"(o), "=&v"(t) : "v"(y0), "v"(y1), "v"(c0), "v"(c1), "s"(sel));
    return o;
}

// The same in two halves for two neighbouring MFMA gaps (round 6: the epilogue of a fragment spread over 16 gaps, two instructions
// each, instead of 8 gaps of four - a tile is bound by the issue of its non-MFMA instructions, and the gaps that held four VALU next
// to a ring read and a counted wait ran over their MFMA's 32 cycles while others stood half empty)
__device__ __forceinline__ void epi2_a(int& o, int& t, const float y0, const float y1, const float c0, const float c1, const uint32_t sel) {
    asm volatile("v_perm_b32 %1, %5, %4, %6\n\t"
                 "v_cvt_pk_f16_f32 %0, %2, %3"
                 : "=&v"(o), "=&v"(t) : "v"(y0), "v"(y1), "v"(c0), "v"(c1), "s"(sel));
}
__device__ __forceinline__ int epi2_b(int o, int t) {
    asm volatile("v_pk_ashrrev_i16 %1, 15, %1 op_sel_hi:[0,1]\n\t"
                 "v_bfi_b32 %0, %1, 0, %0"
                 : "+v"(o), "+v"(t));
    return o;
}

// ---- exact clamp masks (XM): the mask 1[cd >= 0] of zero_clamp comes as one word per (S tile, R position) from a higher-precision
// cd (k_cd_mask) instead of from the sign of the fp16 chain.  The words travel like the tiles: one 256-byte LDS-DMA piece per
// fragment and tile into a four-slot ring three tiles ahead, read back by the lane that owns the R position.
__device__ __forceinline__ void mask_dma(uint32_t lds_dst, uint32_t voff, const char* sbase) {      // 64 dwords -> LDS at lds_dst + 4 lane
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void lds_rd32(uint32_t& d, uint32_t addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr)); }
// word of the lane's R position -> bits of the lane's 16 S positions in accumulator order: element i sits at bit (i & 3) + 8 (i >> 2)
__device__ __forceinline__ void mask_prep(uint32_t& wsh, const uint32_t w, const uint32_t sh) {
    asm volatile("v_lshrrev_b32 %0, %2, %1" : "=v"(wsh) : "v"(w), "v"(sh));
}
// Epilogue of accumulator elements 2J, 2J+1 with the mask bits given: -G = on ? fd'' - shift : 0 as one packed fp16 word, 5 VALU
template <int J> __device__ __forceinline__ int epi2m(const float y0, const float y1, const uint32_t wsh, const uint32_t lo16) {
    constexpr int B0 = 2 * (J & 1) + 8 * (J >> 1);
    int o, m0, m1;
    asm volatile("v_bfe_i32 %1, %5, %c7, 1\n\t"
                 "v_bfe_i32 %2, %5, %c8, 1\n\t"
                 "v_cvt_pk_f16_f32 %0, %3, %4\n\t"
                 "v_bfi_b32 %1, %6, %1, %2\n\t"
                 "v_and_b32 %0, %0, %1"
                 : "=&v"(o), "=&v"(m0), "=&v"(m1) : "v"(y0), "v"(y1), "v"(wsh), "s"(lo16), "n"(B0), "n"(B0 + 1));
    return o;
}

template <int J> __device__ __forceinline__ void epi2m_a(int& o, int& m0, int& m1, const float y0, const float y1, const uint32_t wsh) {
    constexpr int B0 = 2 * (J & 1) + 8 * (J >> 1);
    asm volatile("v_bfe_i32 %1, %5, %c6, 1\n\t"
                 "v_bfe_i32 %2, %5, %c7, 1\n\t"
                 "v_cvt_pk_f16_f32 %0, %3, %4"
                 : "=&v"(o), "=&v"(m0), "=&v"(m1) : "v"(y0), "v"(y1), "v"(wsh), "n"(B0), "n"(B0 + 1));
}
__device__ __forceinline__ int epi2m_b(int o, int m0, int m1, const uint32_t lo16) {
    asm volatile("v_bfi_b32 %1, %3, %1, %2\n\t"
                 "v_and_b32 %0, %0, %1"
                 : "+v"(o), "+v"(m0) : "v"(m1), "s"(lo16));
    return o;
}

// sum over the 64 lanes through DPP row operations (dg_common.h half_sum) instead of six LDS-crossbar shuffles
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define DG_DPP_ADD2(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
    DG_DPP_ADD2(0x111, 0xf); DG_DPP_ADD2(0x112, 0xf); DG_DPP_ADD2(0x114, 0xf); DG_DPP_ADD2(0x118, 0xf);
    DG_DPP_ADD2(0x142, 0xa);
#undef DG_DPP_ADD2
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// NKF feature k-steps (C = 16 NKF), KD = 16 NKD padded code width, NKC code k-steps that are not all padding
// FOLD (round 5): the intra pair-set's streamed-side gradient is formed HERE, with no G tiles and no k_gs job.  Intra streams the
// stationary operand's own image; fd, cd and the clamp mask are symmetric in (p, q), only the centering by ROW means is not:
// -G[p][q] = m (fd + c0 - r_p).  d/dc of both sides is (G + G^T) y with (G + G^T)[p][q] = 2 m (fd + c0 - r_p / 2 - r_q / 2) =: 2 m Z -
// ONE product with the tile the wave holds anyway, if the accumulator holds Z instead of fd + c0 - r_p.  The lane's half (r_p / 2)
// goes into the chain's start value; the streamed position's half rides in the chain as ONE more MFMA: the S tile's unused code
// k-step NKC (channels 16 NKC .. of the zero padding, KD > 16 NKC) carries -r_q / 2 as an fp16 pair (hi, 2048 lo) in k = 0, 1
// (k_rowmean writes it), the B fragment is the constant (1, 2^-11, 0 ...).  The loss sums come out of the gradient tiles
// (sum <x_p, dR_p>) and stay right: sum m cd (r_q - r_p) = 0 by symmetry.  The host doubles the tile's weight (dg_api.hip).
template <int NKF, int NKD, int NKC, bool XM = false, bool DYN = false, bool FOLD = false>      // DYN: the dynamic walk (many items per workgroup), below
__global__ __launch_bounds__(256) void k_corr2(const DgCorrArgs args_k) {
    using BL = BlobT<NKF, NKD>;
    constexpr int RF = 2, NW = 4, KD = BL::KD, NDF = KD / 32, DP = KD;
    // chain steps per fragment and tile: the NKF feature k-steps, the NKC code k-steps - NOT in the exact-mask form (round 6): its mask
    // comes from the words of k_cd_mask3 and G = m (fd'' - shift) needs no cd; the loss and cd sums come out of the gradient tiles and
    // the column sums as ever, so the fp16 cd chain was 10 dead MFMAs of 70 per tile - and FOLD's one extra step
#ifdef C2_XM_KEEP_CD        // (developer A/B: the exact-mask form with its dead cd chain, as until round 5)
    constexpr bool NOCD = false;
#else
    constexpr bool NOCD = XM;
#endif
    constexpr int BUF = BL::BYTES, NS = NKF + (NOCD ? 0 : NKC) + (FOLD ? 1 : 0), PF = C2_PF, NBUF = 4;     // tiles are fetched NBUF - 1 ahead
    static_assert(PF == 8 || PF == 12, "phase C issues the first PF reads of a tile over its six gaps");
    static_assert(!FOLD || NKC < NKD, "FOLD needs a spare code k-step in the blob");
    constexpr int PIECES = BL::CHUNKS / NW;                 // 1-KiB DMA pieces per wave and tile
    constexpr int ADR = RF * NKF * 4;                       // first accumulator register of the gradient accumulators
    static_assert(BL::CHUNKS % NW == 0, "tile chunks must split evenly over the waves");
    static_assert(ADR + RF * 2 * 16 <= 256 && NDF == 3, "accumulator-file plan: Rf + four gradient accumulators");
    // phase-A gaps: 0, 1 mask words, 2..9 the epilogue halves of fragment 1, then the DMA pieces of tile t + 3 - each in two halves over
    // two neighbouring gaps where the chain is long enough (SPLIT_DMA), whole in one gap otherwise (the exact-mask form's short chain)
#ifdef C2_FORCE_SINGLE_DMA  // (developer A/B)
    constexpr bool SPLIT_DMA = false;
#else
    constexpr bool SPLIT_DMA = 10 + 2 * PIECES <= NS;
#endif
    static_assert(10 + PIECES <= NS - 3 && NS >= 22 && NS > PF, "phase-A gaps for the epilogue halves and the DMA pieces / phase-B gaps");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][BUF] tiles, red[8][4]
    declare_agprs();
#ifdef C2_STAMPS
    unsigned long long t_entry;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) :: "memory");
#endif
#ifdef C2_BLOCKLOG          // (finer stamps are kept in registers and stored at the block's end: a store in between would be one more
    unsigned long long bl_t[5] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0};     //  vector-memory operation under the counted waits)
#define BL_T(i) do { bl_t[i] = wall_clock64(); } while (0)
#else
#define BL_T(i) do {} while (0)
#endif

    // ---- PERSISTENT workgroups (round 4): the launch has one workgroup per CU (a multiple of 8: the XCDs are dealt round-robin) and
    //      each walks the work items orig = blockIdx.x, + gridDim.x, ... - what the dispatcher did with one workgroup per item, minus
    //      its 0.8 us between two workgroups of a CU.  All four waves take the same items (wave-uniform control flow).
    const int nitems = args_k.njobs * args_k.B * args_k.nrb + (args_k.gr_list ? args_k.B * args_k.gr_blocks_per_image : 0);
    // The walk is DYNAMIC where a workgroup has many items (six or more on average: config 5's 56 x 56 grid, 12 per workgroup, whose
    // grouped and full items of different lengths a fixed round-robin balances badly: 3.43 -> 3.31 ms): a workgroup's first item is
    // its own index, every further one the next not yet taken on its XCD (one counter per XCD: an item's XCD stays its index modulo
    // 8).  With three or four items per workgroup (the headline) the fixed walk keeps the row blocks that stream the same operand
    // in step and is 0.6 % faster; nothing finer than a 47-us item could make up for a workgroup that starts late either way.  The
    // next index is asked for in front of the item's tile loop (thread 0; the answer is needed 40 us later; asked at the very start
    // of the item the request stood in front of every load of the block's start in the in-order vmcnt queue: +3 % on the kernel)
    // and passed on through LDS at the item's end; an item that turns out to be served by a group asks and waits on the spot.
    __shared__ int next_item_s;
    __shared__ unsigned long long span_keep[2];          // (dg_prof_main_span: thread 0's entry stamps)
    int next_orig = 0;
    if (threadIdx.x == 0) dg_span_enter(args_k.span, span_keep);
    for (int orig = blockIdx.x; orig < nitems; orig = next_orig) {
    unsigned taken = 0;                 // (thread 0) items of this XCD handed out before this request
    // (pointer and mode re-derived per item from the kernel arguments: two scalar registers less across the tile loops)
    constexpr bool dyn = DYN;       // (the launcher's choice: counters present, a grid of whole XCD rounds, six or more items per workgroup)
    auto request = [&]() {
        if (dyn && threadIdx.x == 0) taken = __hip_atomic_fetch_add(&args_k.wctr[blockIdx.x & 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // all threads: the item after this one (dynamic: thread 0's answer through LDS; one barrier)
    auto advance = [&]() {
        if constexpr (DYN) {
            if (threadIdx.x == 0) next_item_s = (int)(blockIdx.x & 7) + 8 * (int)((gridDim.x >> 3) + taken);
            __syncthreads();
            next_orig = next_item_s;
        } else {
            next_orig = orig + (int)gridDim.x;
        }
    };
    // The kernel arguments are read through a pointer hipcc cannot see through, once per item: as loop invariants it kept the job
    // table's fields in scalar registers across the walk, spilled seventy of them into VGPR lanes and took accumulator registers
    // for its own values (the audit of tests/test_host_cpu.py).  Still the kernel-argument segment: scalar loads, as before.
    const DgCorrArgs __attribute__((address_space(4)))* args_p =
        (const DgCorrArgs __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(args_p));
    const DgCorrArgs& args = *(const DgCorrArgs*)args_p;
#ifdef C2_BLOCKLOG
    bl_t[0] = wall_clock64();
#endif
    // ---- XCD-aware item order (as k_corr_main): every XCD owns B/8 whole images; full row blocks first, ragged last
    int bid;
    {
        const int nwg = nitems;
        const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    }
    // (the thread index through an opaque statement, per item: nothing derived from it is carried across items in registers the
    //  tile loop needs - hipcc hoisted all of it out of the walk and parked the overflow in accumulator registers)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int Ppad = args.Ppad, ntiles = Ppad >> 5;
    // What the block works on.  A FULL row block: pair-set jid of image n, row tiles rb*8 .. rb*8+7, one (jid, n) for all
    // eight fragments.  The RAGGED last row block (L = ntiles % 8 row tiles) is cheap in MFMAs but has to stream the whole
    // operand S like a full one: ragged row blocks are therefore GROUPED BY STREAMED OPERAND - all (pair-set, image) whose S is
    // image m of the same operand array (the pair-set's own image for intra, the negatives whose batch map points at m; lists
    // from k_group_ragged) share one block, 8 / L of them at a time, each fragment with its own pair-set and image.  Without
    // lists (or for the consumers beyond a group's blocks: FALLBACK) a ragged block serves one (jid, n) as before.
    int n_first = 0, jid_first = 0, rb = 0, mS = 0;        // (first fragment's (n, jid); mS = image index of the streamed operand)
    const int64_t* sidx_p = nullptr;                        // its batch-map entry (null: mS is final)
    int fj[RF], fn[RF], ft[RF];                             // per fragment of this wave: pair-set, image, row tile (-1: none)
    {
        const int nh = args.njobs;                          // pair-set jobs only (the depth term runs in the k_gs launch)
        const int L = ntiles % (NW * RF);
        const bool ragged = args.nrb > 1 && L != 0;
        const int nfull = args.nrb - (ragged ? 1 : 0);
        const bool grouped = ragged && args.gr_list != nullptr;
        int kind = 0;                                       // 0 full, 1 grouped ragged, 2 ragged of one (jid, n)
        int jid = 0, n = 0, gkey = 0, gpart = 0;
        if ((nitems & 7) == 0 && (args.B & 7) == 0) {
            // every XCD owns B/8 whole images; inside that chunk the long blocks go first: pair-set jobs with a full row block,
            // then the ragged ones
            const int imgs = args.B >> 3;
            const int cA = imgs * nh * nfull;
            const int cG = grouped ? imgs * args.gr_blocks_per_image : 0;
            const int cB = ragged ? imgs * nh : 0;
            const int per_chunk = cA + cG + cB;
            const int xcd = bid / per_chunk;
            int i = bid - xcd * per_chunk, nl;
            // (the grouped ragged blocks are as long as full ones - up to eight fragments - and go FIRST: started last they
            //  would each add a whole block to the makespan; the short one-(pair-set, image) blocks pack the tail)
            if (i < cG) {
                nl = i / args.gr_blocks_per_image; i -= nl * args.gr_blocks_per_image;
                gkey = 0;
                while (i >= args.gr_nblk[gkey]) { i -= args.gr_nblk[gkey]; ++gkey; }
                gpart = i; rb = args.nrb - 1; kind = 1;
            }
            else if (i < cG + cA) { i -= cG; nl = i / (nh * nfull); i -= nl * nh * nfull; jid = i / nfull; rb = i - jid * nfull; kind = 0; }
            else { i -= cA + cG; nl = i / nh; jid = i - nl * nh; rb = args.nrb - 1; kind = 2; }
            n = xcd * imgs + nl;
        } else {
            // (no XCD chunking: grouping is not set up by the launcher for such grids)
            const int per_img = args.njobs * args.nrb;
            n = bid / per_img; bid -= n * per_img; jid = bid / args.nrb; rb = bid - jid * args.nrb;
            kind = ragged && rb == args.nrb - 1 ? 2 : 0;
        }
#ifdef C2_BLOCKLOG          // developer build: per-block timeline (DG_BLOCKLOG=<file>, scripts/blocklog.py)
        if (args.blocklog && threadIdx.x == 0) {
            unsigned long long* e = args.blocklog + (size_t)orig * 16;
            e[0] = __builtin_amdgcn_s_getreg(63492); e[1] = __builtin_amdgcn_s_getreg(63508); e[6] = kind; e[7] = rb;
            e[2] = wall_clock64(); e[3] = e[2]; e[4] = e[2]; e[5] = e[2];
        }
#endif
#pragma unroll
        for (int f = 0; f < RF; ++f) { fj[f] = jid; fn[f] = n; ft[f] = (rb * NW + wid) * RF + f; if (ft[f] >= ntiles) ft[f] = -1; }
        // (the batch-map entry is loaded with the block's other small inputs below and consumed BEHIND the stationary-fragment loads:
        //  as an ordinary load it was a memory round trip of its own in front of everything else - 1.5 of the 5.1 us a block
        //  spent before its first tile, round-4 block log)
        mS = n; sidx_p = args.jobs[jid].sidx ? args.jobs[jid].sidx + n : nullptr;
        jid_first = jid; n_first = n;
        if (kind == 2 && grouped) {
            // this (jid, n) is served by a grouped block unless its rank lies beyond the group's blocks
            const int key = args.gr_key[jid];
            if ((int)args.gr_rank[jid * args.B + n] < args.gr_nblk[key] * args.gr_cpb) { request(); advance(); continue; }
        }
        if (kind == 1) {
            // image n of this chunk is the STREAMED image; the consumers come from the list of (key, n)
            mS = n; sidx_p = nullptr;
            const int cnt = args.gr_count[gkey * args.B + n];
            const int c0 = gpart * args.gr_cpb;
            if (c0 >= cnt) { request(); advance(); continue; }
            jid_first = args.gr_first[gkey];
            bool any = false;
#pragma unroll
            for (int f = 0; f < RF; ++f) {
                const int slot = wid * RF + f, ci = c0 + slot / L;
                ft[f] = -1;
                if (slot / L < args.gr_cpb && ci < cnt) {
                    const int e = args.gr_list[(gkey * args.B + n) * DG_GR_CAP + ci];
                    fj[f] = e & 255; fn[f] = e >> 8; ft[f] = (args.nrb - 1) * (NW * RF) + slot % L;
                    any = true;
                }
            }
            (void)any;
            n_first = -1;
        }
    }
    const DgJob& job = args.jobs[jid_first];                // (of a grouped block: the group's first pair-set - same S array, Scsum)

    // ---- the two 32-row tiles of R owned by this wave
    bool act[RF];
    int pr[RF];
    const char* Rblob[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) {
        act[f] = ft[f] >= 0;                                            // wave-uniform
        fj[f] = __builtin_amdgcn_readfirstlane(fj[f]); fn[f] = __builtin_amdgcn_readfirstlane(fn[f]); ft[f] = __builtin_amdgcn_readfirstlane(ft[f]);
        pr[f] = act[f] ? ft[f] * 32 + r : 0;
        Rblob[f] = args.jobs[fj[f]].Rop + ((size_t)fn[f] * ntiles + (act[f] ? ft[f] : 0)) * BL::BYTES;
    }
    // (the block end's output pointers, fetched with the other job fields: as kernel-argument loads at the block end each was a
    //  scalar-memory round trip of its own with nothing else to do - round-4 block log: 1.3 us behind the block barrier)
    float* dRp[RF];
    float* partp[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) { dRp[f] = args.jobs[fj[f]].dR; partp[f] = args.jobs[fj[f]].part; }
    (void)n_first;
    bool fo[RF];                                          // FOLD: this fragment's pair-set is folded (wave-uniform)
#pragma unroll
    for (int f = 0; f < RF; ++f) fo[f] = FOLD && args.jobs[fj[f]].fold != 0;

    // ---- small per-block inputs first (their latency runs under the 270 KB that follow): the B per-image sums of the row means
    //      (m0), this lane's row means, the streamed operand's code column sums.  asm loads: hipcc would wait for a load it knows
    //      with vmcnt(0) at its first use, i.e. for every DMA piece issued since; they are consumed behind the first counted wait.
    float rimg_v[RF] = {0.f, 0.f}, rvec_v[RF] = {0.f, 0.f}, cs_pre[NDF];
    int mS_ld = 0;
    const float* const zsrc = reinterpret_cast<const float*>(args.dummy);          // any valid address
    {
        const void* p_sidx = sidx_p ? reinterpret_cast<const void*>(sidx_p) : reinterpret_cast<const void*>(zsrc);
        asm volatile("global_load_dword %0, %1, off" : "=v"(mS_ld) : "v"(p_sidx) : "memory");      // (low word of the int64 entry)
#pragma unroll
        for (int f = 0; f < RF; ++f) {
            const DgJob& jf = args.jobs[fj[f]];
            const float* p_rimg = jf.rvec ? jf.rimg + (lane < args.B ? lane : 0) : zsrc;
            asm volatile("global_load_dword %0, %1, off" : "=v"(rimg_v[f]) : "v"(p_rimg) : "memory");
            const float* p_rv = jf.rvec ? jf.rvec + (size_t)fn[f] * Ppad + pr[f] : zsrc;
            asm volatile("global_load_dword %0, %1, off" : "=v"(rvec_v[f]) : "v"(p_rv) : "memory");
        }
    }

    // ---- stationary operand: feature fragments -> accumulator registers, code fragments -> arch VGPRs
    const uint32_t smem_a = lds_addr(smem);
    const int sw = (r >> 2) & 3;
    const int fb0 = r * 64 + ((h ^ sw) * 16), fb1 = r * 64 + (((2 + h) ^ sw) * 16);        // dg_f_off(r, 2 ks + h), ks even / odd
    // (fragment 0 only: fragment 1's 24 KB per wave are first used in phase B of tile 0 and are issued BEHIND the pieces of tile 0 -
    //  round 4: the block's first tile waited for all 67 KB per wave, 15.5 k cycles of a 98-k-cycle block)
    auto load_rfrag = [&](auto F) {
        sfor<NKF>([&](auto KS) {
            constexpr int f = F.value, ks = KS.value;
            agpr_load16<f * NKF + ks>(Rblob[f] + ((ks & 1) ? fb1 : fb0) + (ks >> 1) * 2048);
        });
    };
    load_rfrag(std::integral_constant<int, 0>{});
    // (NOCD - the exact-mask form, which has no cd chain: these fragments are needed at the block end only, and are loaded THERE by
    //  ordinary loads.  As asm loads up here with no reader inside the loop, hipcc - for which an asm statement's output exists the
    //  moment the statement ends - copied them to other registers before they had landed: NaN loss sums, round 6.)
    v4i_t Rc[RF][NKC];                                    // B operands of the cd chain: granule 2k + h of row r
    const char* rc_lane[RF];                              // (NOCD: this lane's address of granule h, kept in a vector register pair - as
                                                          //  scalar pointers across the tile loop they spilled into VGPR lanes)
#pragma unroll
    for (int f = 0; f < RF; ++f) { rc_lane[f] = Rblob[f] + BL::OFF_C + (h * 32 + r) * 16; if constexpr (NOCD) asm volatile("" : "+v"(rc_lane[f])); }
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int k = 0; k < (NOCD ? 0 : NKC); ++k)
            // (asm: a load hipcc knows about would make it wait for vmcnt at the first use INSIDE the tile loop - on every
            //  iteration, draining the DMA pipeline; these complete before the first tile's counted wait: they are older)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(Rc[f][k]) : "v"(Rblob[f] + BL::OFF_C + ((2 * k + h) * 32 + r) * 16) : "memory");
    sfor<RF * 2 * 4>([&](auto I) { agpr_zero4<ADR + 4 * I.value>(); });
    // the streamed image: its batch-map entry has landed (it is the oldest load; younger: 2 RF small inputs, NKF + RF NKC fragment loads)
    static_assert(2 * RF + NKF + RF * NKC == 38, "literal wait counts below");
    if constexpr (NOCD) asm volatile("s_waitcnt vmcnt(28)" : "+v"(mS_ld) :: "memory");       // (no code fragment loads)
    else asm volatile("s_waitcnt vmcnt(38)" : "+v"(mS_ld) :: "memory");
    const int nS = __builtin_amdgcn_readfirstlane(sidx_p ? mS_ld : mS);
#pragma unroll
    for (int d = 0; d < NDF; ++d) {
        const float* p_cs = job.Scsum ? job.Scsum + (size_t)nS * KD + 32 * d + r : zsrc;
        asm volatile("global_load_dword %0, %1, off" : "=v"(cs_pre[d]) : "v"(p_cs) : "memory");
    }
    acc_t dRv[RF];                                        // gradient accumulator of channel group 2 (groups 0, 1: accumulator file)
#pragma unroll
    for (int f = 0; f < RF; ++f) dRv[f] = acc_t{};

    // ---- XM: mask words of this wave's fragments, [wave][fragment][4 slots][64 dwords] behind the tile buffers.  The word of S tile
    //      T is fetched three tiles ahead (fragment 1, whose epilogue runs one phase later: two), always IN FRONT of the tile pieces
    //      issued in the same iteration, so the counted wait that covers those pieces covers it
    const uint32_t mring_a = smem_a + NBUF * BUF + C2_RED_BYTES;
    const char* mbase[RF] = {nullptr, nullptr};
    uint32_t mvoff[RF] = {0u, 0u};
    auto mask_issue = [&](const int f, const int T) {
        const int Tc = T < ntiles ? T : 0;               // past the end: a dummy piece keeps the counted waits uniform
        mask_dma(mring_a + ((wid * RF + f) * 4 + (T & 3)) * 256, mvoff[f], mbase[f] + (size_t)Tc * Ppad * 4);
    };
    if constexpr (XM) {
#pragma unroll
        for (int f = 0; f < RF; ++f) {
            mbase[f] = reinterpret_cast<const char*>(args.jobs[fj[f]].maskbits) + (size_t)fn[f] * ntiles * Ppad * 4;
            mvoff[f] = (uint32_t)pr[f] * 4u;
        }
        if (act[0]) { mask_issue(0, 0); mask_issue(0, 1); mask_issue(0, 2); }
        if (act[1]) { mask_issue(1, 0); mask_issue(1, 1); }
    }

    // ---- tile staging: chunk c of a tile is fetched by wave c % 4 (piece k of wave w = chunk w + 4 k)
    const char* const Sop_img = job.Sop + (size_t)nS * ntiles * BL::BYTES;    // wave-uniform
    const uint32_t dma_voff = lane * 16 + wid * 1024;
    auto issue_tile_piece = [&](auto K, int t, int b) {
        const int tt = t < ntiles ? t : 0;                // past the end: a dummy piece keeps the counted waits uniform
        const char* sb = Sop_img + (size_t)tt * BL::BYTES;
        const uint32_t dst = smem_a + b * BUF + wid * 1024;
        dma_piece<K.value * 4096>(dst, dma_voff, sb);
    };
    sfor<PIECES>([&](auto K) { issue_tile_piece(K, 0, 0); });
    static_assert(RF == 2, "fragment 1 behind tile 0");
    load_rfrag(std::integral_constant<int, 1>{});         // a[4 NKF ..]: landed at the counted wait in front of phase B (tile 0)
    sfor<PIECES>([&](auto K) { issue_tile_piece(K, 1, 1); });
    sfor<PIECES>([&](auto K) { issue_tile_piece(K, 2, 2); });

    // (per-job scalars: computed behind the first counted wait, from the values loaded at the top of the block)
    // per-lane LDS byte addresses of the fragments of the current tile
    const int crow = (h * 32 + r) * 16;
    uint32_t va0 = smem_a + fb0, va1 = smem_a + fb1, vc = smem_a + BL::OFF_C + crow, vp = smem_a + BL::OFF_P + (h * KD + r) * 16;
    auto rd_step = [&](auto ST, v4i_t& d) {               // A fragment of chain step ST (feature k-steps, then code k-steps)
        constexpr int st = ST.value;
        if constexpr (st < NKF) { if constexpr (st & 1) lds_rd<(st >> 1) * 2048>(d, va1); else lds_rd<(st >> 1) * 2048>(d, va0); }
        else lds_rd<(NOCD ? NKC : st - NKF) * 1024>(d, vc);          // (XM: the only step behind the feature steps is FOLD's, code k-step NKC)
    };

    acc_t Yf[RF], Yc[RF];
    v4i_t ga[RF][2];                                     // -G as fp16 A fragments: k-step sp holds accumulator elements 8sp..8sp+7
    v4i_t ra[PF], bP[2 * NDF];

    BL_T(1);
    // tile 0 landed (everything older - the small inputs, the fragment loads of fragment 0, the code fragments - is complete as
    // well); younger: fragment 1's NKF loads, the pieces of tiles 1 and 2
    static_assert(2 * PIECES + NKF == 42, "literal wait count below");
    asm volatile("s_waitcnt vmcnt(42)" : "+v"(rimg_v[0]), "+v"(rimg_v[1]), "+v"(rvec_v[0]), "+v"(rvec_v[1]), "+v"(cs_pre[0]), "+v"(cs_pre[1]), "+v"(cs_pre[2]) :: "memory");
    BL_T(2);
    // per-fragment scalars: fd'' - shift = Yf - rowmean + (m0 - shift); the chain starts at c0_lane
    double c0pair[RF];                                    // (c0_lane, c0_lane): source of the v_mov_b64 accumulator initialisation
    {
        if (!job.Scsum) { cs_pre[0] = cs_pre[1] = cs_pre[2] = 0.f; }
#pragma unroll
        for (int f = 0; f < RF; ++f) {
            const DgJob& jf = args.jobs[fj[f]];
            float c0 = -jf.shift;
            if (jf.rvec) c0 += wave_sum_dpp(lane < args.B ? rimg_v[f] : 0.f) * args.inv_BP;      // (B <= 64 images per call of this form)
            const float cl = jf.rvec ? c0 - (fo[f] ? 0.5f * rvec_v[f] : rvec_v[f]) : c0;
            const float2 two = make_float2(cl, cl);
            c0pair[f] = __builtin_bit_cast(double, two);
            asm volatile("" : "+v"(c0pair[f]));
        }
    }
    acc_t c0splat[RF];                                    // C operand of the first MFMA of every fd chain
#pragma unroll
    for (int f = 0; f < RF; ++f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { Yf[f][i] = c0pair[f]; c0splat[f][i] = c0pair[f]; }
        asm volatile("" : "+v"(c0splat[f]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // XM: mwA = word of fragment 0 for the tile whose epilogue comes next (phase B), mwB = word of fragment 1 for the tile before
    // (its epilogue runs in phase A of the following tile); both are read in phase C, in FRONT of the first fragment reads of the
    // next tile, and have landed once the first counted lgkmcnt wait of phase A has passed.  wsh: the word shifted to the lane's half
    uint32_t mwA = 0u, mwB = 0u, wsh[RF] = {0u, 0u};
    const uint32_t msh = 4u * (uint32_t)h;
    const uint32_t mrd_a = mring_a + wid * (RF * 4 * 256) + lane * 4;
    const uint32_t lo16 = __builtin_amdgcn_readfirstlane(0x0000ffff);
    if constexpr (XM) { if (act[0]) lds_rd32(mwA, mrd_a); }
    sfor<PF>([&](auto I) { rd_step(I, ra[I.value]); });

    const uint32_t perm_sel = __builtin_amdgcn_readfirstlane(0x07060302);      // {c1 bytes 3,2 | c0 bytes 3,2}
    auto epi_pair = [&](auto F, auto J) {                  // accumulator elements 2j, 2j+1 of fragment f -> one fp16 word of -G
        constexpr int f = F.value, j = J.value;
        const f32x16 yf = __builtin_bit_cast(f32x16, Yf[f]);
        auto& YcL = Yc;                      // (named outside the asm operand below: clang captures it for the generic lambda only so)
        if constexpr (XM) {
            ga[f][j >> 2][j & 3] = epi2m<j>(yf[2 * j], yf[2 * j + 1], wsh[f], lo16);
            (void)YcL;                       // (no cd chain in this form: the mask comes from the words)
        } else {
            const f32x16 yc = __builtin_bit_cast(f32x16, Yc[f]);
            ga[f][j >> 2][j & 3] = epi2(yf[2 * j], yf[2 * j + 1], yc[2 * j], yc[2 * j + 1], perm_sel);
        }
    };
    // half HH (0..15) of the epilogue of fragment f: pair HH >> 1, first half (mask source + fp16 pack) or second (mask applied)
    int eo[RF] = {0, 0}, et[RF] = {0, 0}, eu[RF] = {0, 0};
    auto epi_half = [&](auto F, auto HHc) {
        constexpr int f = F.value, HH = HHc.value, j = HH >> 1;
        if constexpr ((HH & 1) == 0) {
            const f32x16 yf = __builtin_bit_cast(f32x16, Yf[f]);
            if constexpr (XM) epi2m_a<j>(eo[f], et[f], eu[f], yf[2 * j], yf[2 * j + 1], wsh[f]);
            else {
                const f32x16 yc = __builtin_bit_cast(f32x16, Yc[f]);
                epi2_a(eo[f], et[f], yf[2 * j], yf[2 * j + 1], yc[2 * j], yc[2 * j + 1], perm_sel);
            }
        } else {
            if constexpr (XM) ga[f][j >> 2][j & 3] = epi2m_b(eo[f], et[f], eu[f], lo16);
            else ga[f][j >> 2][j & 3] = epi2_b(eo[f], et[f]);
        }
    };
    (void)epi_pair;
    // per-fragment base of the G tiles [image][S tile t][R tile]: resolved HERE - a kernel-argument (scalar) load inside the
    // tile loop would have to be waited for with lgkmcnt(0), i.e. together with every LDS read in flight
    v4i_t* gbase[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f)
#ifdef C2_ABL_GSTORE      // (developer ablation, WRONG results: every G store of the launch into the same 2 KiB - what do the 246 MB of G writes cost the loop?)
        gbase[f] = reinterpret_cast<v4i_t*>(args.jobs[fj[f]].Gout) + lane;
#else
        gbase[f] = reinterpret_cast<v4i_t*>(args.jobs[fj[f]].Gout) + (fo[f] ? (size_t)0 : ((size_t)fn[f] * ntiles * ntiles + (act[f] ? ft[f] : 0)) * 128) + lane;
#endif
    const size_t gstep = (size_t)ntiles * 128;
    // (a folded fragment has no reader for its G tiles; its stores stay in the instruction stream - the counted vmcnt waits of the
    //  tile barrier count them - but all go to the first 2 KiB of the pair-set's buffer with the default cache policy: L2 traffic)
    size_t gstepb[RF];
#pragma unroll
#ifdef C2_ABL_GSTORE
    for (int f = 0; f < RF; ++f) gstepb[f] = 0;
#else
    for (int f = 0; f < RF; ++f) gstepb[f] = fo[f] ? (size_t)0 : gstep * sizeof(v4i_t);
#endif
    // running store addresses: a scalar base per fragment (advanced by scalar adds; the tile index times the tile stride as 64-bit scalar
    // multiplies in front of every store cost three s_mul and two adds each) + ONE 32-bit lane offset for every store - half the address
    // bytes of the vaddr form and no 64-bit VALU add (round 6: -0.5 % of the kernel, -2 us of the step).  gsb[0] points at S tile t,
    // gsb[1] at S tile t - 1 (fragment 1 runs one phase behind).  (dg_uniform_ptr: readfirstlane returns int - the halves go through
    // uint32_t, or the low one is sign-extended over the high one)
    uint64_t gsb[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f)
        gsb[f] = reinterpret_cast<uint64_t>(dg_uniform_ptr(reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(gbase[f] - lane) - (f == 1 ? gstepb[1] : 0))));
    const uint32_t goff = lane * 16;
    auto g_store = [&](const int f, const int sp, int t) {
        // (asm: the store must be ISSUED here - the counted vmcnt waits at the tile barrier rely on it; hipcc is free to sink
        //  an ordinary store past the barrier, after which the wait lets the youngest DMA pieces of the next tile slip)
        (void)t;
#ifdef C2_ABL_NOGSTORE     // (developer ablation, WRONG results: no G store is issued)
        return;
#endif
        // (non-temporal: with the default cache policy on these stores / k_gs's loads the step is 1-6 % slower, profiles/r03_SUMMARY.md)
        if (sp == 0) {
            if (FOLD && __builtin_expect(fo[f], 0)) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(goff), "v"(ga[f][sp]), "s"(gsb[f]) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(goff), "v"(ga[f][sp]), "s"(gsb[f]) : "memory");
        } else {
            if (FOLD && __builtin_expect(fo[f], 0)) asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024" :: "v"(goff), "v"(ga[f][sp]), "s"(gsb[f]) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024 nt" :: "v"(goff), "v"(ga[f][sp]), "s"(gsb[f]) : "memory");
        }
    };
    // FOLD: the B fragment of the extra chain step - k = 0: 1, k = 1: 2^-11 (lanes of half 0 hold k 0..7), zero for a fragment that is
    // not folded.  It borrows ga[f][0]: free between the gradient product that read it and the next epilogue of the fragment.
    const int fold_w = h == 0 ? 0x10003C00 : 0;


#ifdef C2_STAMPS       // developer build: cycle stamps of one block's tile loop (make EXTRA="-DDG_DEVTOOLS -DC2_STAMPS", DG_STAMPS=<file>)
    uint32_t* const st_lds = reinterpret_cast<uint32_t*>(smem + NBUF * BUF + C2_RED_BYTES + (XM ? 4 * 2 * 4 * 256 : 0));      // (XM: behind the mask-word ring)
    #ifndef C2_STAMP_N
#define C2_STAMP_N 0
#define C2_STAMP_J 0
#define C2_STAMP_RB 0
#endif
    const bool stamping = args.stamps != nullptr && n_first == C2_STAMP_N && jid_first == C2_STAMP_J && rb == C2_STAMP_RB;
    auto STAMP = [&](int t, int k) {
#ifndef C2_BLOCKSTAMPS_ONLY
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
        if (stamping && lane == 0 && t < 25) st_lds[(wid * 25 + t) * 6 + k] = (uint32_t)tm;
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    auto BSTAMP = [&](int k) {       // block-level: 0 kernel entry (taken at the top), 1 first tile landed, 2 loop done, 3 block done
        unsigned long long tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
        if (stamping && lane == 0) st_lds[NW * 25 * 6 + wid * 4 + k] = (uint32_t)tm;
    };
    if (stamping && lane == 0) st_lds[NW * 25 * 6 + wid * 4] = (uint32_t)t_entry;
    BSTAMP(1);
#else
    auto STAMP = [&](int, int) {};
    auto BSTAMP = [&](int) {};
#endif
    // The tile loop in three forms: both fragments active, fragment 0 only (ragged last row block), fetch-only wave.
    // LDS reads of a tile are numbered idx = 0 .. TOT-1 (chain of fragment 0, then of fragment 1: the same NS addresses again);
    // read idx lands in ring register idx % PF and is issued right behind MFMA idx - PF (the first PF: behind the previous
    // tile's barrier).  Every second MFMA waits for its own and the next fragment with a counted lgkmcnt.
    // Schedule of tile t (both fragments): fragment 1 runs one phase behind, so that BOTH epilogues sit in the gaps of a long chain
    //     A   chain of fragment 0 (t)            gaps: epilogue of fragment 1 (t-1), its G store, the DMA pieces of tile t+2
    //     A'  dR_1 += G_1^T ScP (t-1), 6 MFMAs   gaps: fd initialisation of fragment 1
    //     B   chain of fragment 1 (t)            gaps: epilogue of fragment 0 (t), its G stores, the gradient B fragments of tile t
    //         -- counted vmcnt + the one workgroup barrier of the tile --
    //     C   dR_0 += G_0^T ScP (t), 6 MFMAs     gaps: first fragments of tile t+1, fd initialisation of fragment 0
    // (tile 0: the "previous" fragment-1 state is all-zero G and zero B fragments; only its G store is skipped)
    // (FOLD: every chain of the launch runs the extra step - with a zero B fragment where the fragment is not folded.  Two forms of
    //  the tile loop, chosen per block, were tried: hipcc then parks values in accumulator registers at the join - the audit of
    //  tests/test_host_cpu.py fails - for 4 us of MFMA time.)
    auto run = [&](auto A0, auto A1) {
        constexpr bool ACT0 = A0.value, ACT1 = A1.value, FRUN = FOLD;
        constexpr int TOT = ACT1 ? 2 * NS : NS;
        constexpr int BP0 = ACT1 ? 2 * NS - 9 : TOT;            // first MFMA index whose gap carries a gradient-B read (phase B gaps NS-9 ..)
        constexpr int MM = XM ? (ACT0 ? 1 : 0) + (ACT1 ? 1 : 0) : 0;       // mask-word pieces per tile
        auto epi1_half = [&](auto HH) { epi_half(std::integral_constant<int, 1>{}, HH); };   // half HH (0..15) of the epilogue of fragment 1
        auto dr1 = [&](auto Q) {
            constexpr int q = Q.value, sp = q / NDF, d = q % NDF;
            if constexpr (d < 2) mfma_h_acc<ADR + 32 + d * 16>(ga[1][sp], bP[q]); else mfma_h8(dRv[1], ga[1][sp], bP[q]);
        };
        int bcur = 0;
        uint32_t dtmp = 0;
        for (int t = 0; t < ntiles; ++t) {
            const int bnext = (bcur + 1) & 3, bprev = (bcur + 3) & 3;
            auto chain_slot = [&](auto IDX, auto F) {          // wait (every second step), MFMA idx, refill of its ring register
                constexpr int idx = IDX.value, f = F.value, st = idx - f * NS;
                if constexpr ((idx & 1) == 0) {
                    constexpr int need = idx + 1 < TOT ? idx + 1 : TOT - 1;
                    constexpr int issued = (idx + PF < TOT ? idx + PF : TOT) + (idx > BP0 ? (idx - BP0 < 2 * NDF ? idx - BP0 : 2 * NDF) : 0);
                    wait_lgkm<issued - (need + 1)>();
                }
                if constexpr (st == 0) mfma_fd8_from<f * NKF>(Yf[f], ra[idx % PF], c0splat[f]);
                else if constexpr (st < NKF) mfma_fd8<f * NKF + st>(Yf[f], ra[idx % PF]);
                else if constexpr (FRUN && st == NS - 1) mfma_h8(Yf[f], ra[idx % PF], ga[f][0]);      // Z: the streamed position's half of the centering
                else if constexpr (st == NKF) mfma_h80(Yc[f], ra[idx % PF], Rc[f][0]);
                else mfma_h8(Yc[f], ra[idx % PF], Rc[f][st - NKF]);
                if constexpr (idx + PF < TOT) rd_step(std::integral_constant<int, (idx + PF) % NS>{}, ra[idx % PF]);
            };
            // DMA source / destination of tile t+3 (scalar)
#ifdef C2_ABL_DMA         // (developer ablation, WRONG results: every tile of the loop is tile 0 of the streamed image - what does the arrival of the tiles cost?)
            const int t2 = 0;
#else
            const int t2 = t + 3 < ntiles ? t + 3 : 0;            // past the end: dummy pieces keep the counted waits uniform
#endif
            const char* const sb2 = Sop_img + (size_t)t2 * BL::BYTES;
            const uint32_t dst2 = smem_a + bprev * BUF + wid * 1024;
            STAMP(t, 0);
            // ================= phase A: Y chain of fragment 0 =================
            sfor<NS>([&](auto ST) {
                constexpr int st = ST.value;
                if constexpr (ACT0) chain_slot(ST, std::integral_constant<int, 0>{});
                if constexpr (XM && ACT0 && st == 0) mask_issue(0, t + 3);
                if constexpr (XM && ACT1 && st == 1) { mask_issue(1, t + 2); mask_prep(wsh[1], mwB, msh); }
                if constexpr (ACT1 && st >= 2 && st < 18) epi1_half(std::integral_constant<int, st - 2>{});
#ifndef C2_ABL_NODMA       // (developer ablation, WRONG results: no tile is fetched inside the loop - what does ISSUING the nine pieces cost?)
                if constexpr (SPLIT_DMA && st >= 10 && st < 10 + 2 * PIECES) {           // piece k: M0 + offset in gap 10 + 2k, the load in gap 11 + 2k
                    constexpr int k = (st - 10) / 2;
                    if constexpr (((st - 10) & 1) == 0) dma_setup<k * 4096>(dst2, dma_voff, dtmp); else dma_go(dtmp, sb2);
                }
                if constexpr (!SPLIT_DMA && st >= 10 && st < 10 + PIECES) dma_piece<(st - 10) * 4096>(dst2, dma_voff, sb2);
#endif
                if constexpr (FRUN && ACT0 && st == NS - 3) ga[0][0] = v4i_t{fo[0] ? fold_w : 0, 0, 0, 0};
                if constexpr (ACT1 && st == NS - 1) { if (t > 0) g_store(1, 0, t - 1); }
                __builtin_amdgcn_sched_barrier(0);
            });
            STAMP(t, 1);
            // ================= phase A': dR_1 += G_1^T ScP of tile t-1 =================
            if constexpr (ACT1) {
                sfor<2 * NDF>([&](auto Q) {
                    constexpr int q = Q.value;
                    dr1(Q);
                    if constexpr (q == 0) { if (t > 0) g_store(1, 1, t - 1); }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // ================= phase B: Y chain of fragment 1, epilogue of fragment 0 in its gaps =================
            // Tile 0: fragment 1's stationary fragments (issued behind the pieces of tile 0) have landed - younger than the last
            // of them are the pieces of tiles 1 and 2 and what phase A issued (9 pieces, XM: its mask words; no G store at tile
            // 0).  Every later tile: weaker than what the tile barrier already asked for one phase ago, never waits.
            static_assert(3 * PIECES == 27, "literal wait count below");
            asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
            sfor<NS>([&](auto ST) {
                constexpr int st = ST.value;
                if constexpr (ACT1) chain_slot(std::integral_constant<int, NS + st>{}, std::integral_constant<int, 1>{});
                // (fragment 0 only: no MFMAs of a second chain stand between the chain that wrote the accumulators and the
                //  epilogue that reads them - wait the chain's last MFMAs out explicitly)
                if constexpr (ACT0 && !ACT1 && st == 0) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
                if constexpr (ACT0) {
                    if constexpr (XM && st == 1) mask_prep(wsh[0], mwA, msh);
                    if constexpr (st >= 2 && st < 18) epi_half(std::integral_constant<int, 0>{}, std::integral_constant<int, st - 2>{});
                    if constexpr (st == 11) g_store(0, 0, t);             // (pairs 0..3 are complete behind gap 9)
                    if constexpr (st == 19) g_store(0, 1, t);             // (pairs 4..7 behind gap 17)
                    if constexpr (FRUN && ACT1 && st == 16) ga[1][0] = v4i_t{fo[1] ? fold_w : 0, 0, 0, 0};
                    if constexpr (st >= NS - 9 && st < NS - 9 + 2 * NDF) {      // B fragments of the gradient products (shared by both fragments)
                        constexpr int q = st - (NS - 9), sp = q / NDF, d = q % NDF;
                        lds_rd<d * 512 + sp * (2 * KD * 16)>(bP[q], vp);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            STAMP(t, 2);
            // ---- tile t+1 landed, every wave is done reading tile t: one barrier per tile.  Younger than the last piece of tile
            //      t+1 (issued in phase A of t-2): per later tile 2 G stores of each fragment and the 9 pieces of tiles t+2, t+3
            {
                constexpr int S0 = ACT0 ? 2 : 0, S1 = ACT1 ? 2 : 0;
                constexpr int PL = PIECES;
                // two tiles' pieces and up to three tiles' G stores are younger than the last piece of tile t+1.  From tile 3 on
                // the count is a constant: that path must not pass through the switch below - hipcc lowers it to a chain of
                // compare-and-branch blocks of which the steady state took three TAKEN branches per tile, 40-80 cycles each on one
                // wave per SIMD (round-4 stamps: a fixed 240-cycle "wait" in front of a barrier all four waves reach within 50)
                // (XM: per tile MM mask-word pieces, issued in front of the tile pieces of their iteration - one iteration's worth
                //  is younger than the last piece of tile t+1 at tile 0, two from then on)
                if (__builtin_expect(t >= 3, 1)) {
                    asm volatile("s_waitcnt vmcnt(%c0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(2 * PL + 3 * S0 + 3 * S1 + 2 * MM) : "memory");
                } else {
                    const int tc0 = t < 2 ? t + 1 : 3, tc1 = t;
                    wait_vm_lgkm_barrier(2 * PL + S0 * tc0 + S1 * tc1 + MM * (t == 0 ? 1 : 2));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            STAMP(t, 3);
            // fragment addresses of tile t+1
            va0 = smem_a + bnext * BUF + fb0; va1 = smem_a + bnext * BUF + fb1;
            vc = smem_a + bnext * BUF + BL::OFF_C + crow; vp = smem_a + bnext * BUF + BL::OFF_P + (h * KD + r) * 16;
            // ================= phase C: dR_0 += G_0^T ScP =================
            sfor<2 * NDF>([&](auto Q) {
                constexpr int qm = Q.value, q = Q.value;
                constexpr int sp = qm / NDF, d = qm % NDF;
                if constexpr (ACT0) {
                    if constexpr (d < 2) mfma_h_acc<ADR + d * 16>(ga[0][sp], bP[qm]); else mfma_h8(dRv[0], ga[0][sp], bP[qm]);
                    if constexpr (XM && q == 0) {          // landed behind this tile's barrier; in front of the fragment reads
                        if constexpr (ACT1) lds_rd32(mwB, mrd_a + 4 * 256 + (uint32_t)(t & 3) * 256);
                        lds_rd32(mwA, mrd_a + (uint32_t)((t + 1) & 3) * 256);
                    }
                    // first PF fragments of tile t+1 (valid after the barrier), issued in index order (the counted waits rely on it)
                    if constexpr (PF == 12 || q < 2) { rd_step(std::integral_constant<int, 2 * q>{}, ra[2 * q]); rd_step(std::integral_constant<int, 2 * q + 1>{}, ra[2 * q + 1]); }
                    else rd_step(std::integral_constant<int, q + 2>{}, ra[q + 2]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            STAMP(t, 4);
            STAMP(t, 5);
            bcur = bnext;
#pragma unroll
            for (int f = 0; f < RF; ++f) { gsb[f] += gstepb[f]; asm volatile("" : "+s"(gsb[f])); }
        }
        // ---- tail: fragment 1 of the last tile (epilogue, G store, gradient product)
        if constexpr (ACT1) {
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            if constexpr (XM) { wait_lgkm<0>(); mask_prep(wsh[1], mwB, msh); }
            sfor<16>([&](auto HH) { epi1_half(HH); });
            g_store(1, 0, ntiles - 1); g_store(1, 1, ntiles - 1);
            asm volatile("s_nop 1" ::: "memory");
            sfor<2 * NDF>([&](auto Q) { dr1(Q); });
        }
        // The gradient accumulators of channel group 2 live in architectural VGPRs and the MFMAs that write them are inline asm:
        // hipcc does not know that their results need 18 wait states before a VALU instruction may read them.  Where the three
        // forms of this loop join it copies half of dRv[1] to other registers RIGHT BEHIND the last MFMA of the tail (one s_nop 0
        // in between) and so dropped that MFMA's contribution - positions 16..31 of the last streamed tile, channels 64..95 - from
        // the odd-numbered rows of every second row tile.  Invisible wherever the last tile is at least half padding (P = 784: 16
        // of its 32 positions), a 1e-4 bias of the loss sums and 2 % errors in those gradient elements on grids with P a multiple
        // of 32 (config 5's 56 x 56; found in round 4).  The wait states are spent HERE, with the accumulators as operands of the
        // statement, so that every compiler-generated read comes behind them.
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(dRv[0]), "+v"(dRv[1]));
    };
    // "previous tile" state of fragment 1 in front of tile 0: cd = -1 everywhere (mask off, G = 0), zero gradient B fragments
    // (XM: the word of that state, mwB, starts as zero - every element off)
    if constexpr (!XM) {
#pragma unroll
        for (int i = 0; i < 8; ++i) Yc[1][i] = __builtin_bit_cast(double, make_float2(-1.f, -1.f));
        asm volatile("" : "+v"(Yc[1]));
    }
#pragma unroll
    for (int q = 0; q < 2 * NDF; ++q) bP[q] = v4i_t{0, 0, 0, 0};
#ifdef C2_BLOCKLOG
    // (e[13], e[14]: the shader clock - s_memtime - at the two ends of the tile loop, next to the 100-MHz wall clock in e[3], e[4]:
    //  cycles / wall time = the clock this CU HELD while it ran the loop, scripts/held_clock.py)
    if (args.blocklog && threadIdx.x == 0) {
        unsigned long long* e = args.blocklog + (size_t)orig * 16; unsigned long long tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
        e[3] = wall_clock64(); e[13] = tm;
    }
#endif
    request();
    if (act[1]) run(std::true_type{}, std::true_type{});
    else if (act[0]) run(std::true_type{}, std::false_type{});
    else run(std::false_type{}, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // no LDS-DMA piece may outlive the workgroup's LDS allocation
    BSTAMP(2);
#ifdef C2_BLOCKLOG
    if (args.blocklog && threadIdx.x == 0) {
        unsigned long long* e = args.blocklog + (size_t)orig * 16; unsigned long long tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
        e[4] = wall_clock64(); e[14] = tm;
    }
#endif

    // ---- block end: raw gradient tiles (accumulator order, as k_corr_main) and the block's partial sums
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last gradient MFMAs have retired before their registers are read
    if constexpr (NOCD) {                                       // (requested first: the gradient tiles' reads and stores run under their latency)
#pragma unroll
        for (int f = 0; f < RF; ++f)
#pragma unroll
            for (int k = 0; k < NKC; ++k)
                // (a GLOBAL-address-space load: through the generic pointer hipcc emitted flat_load_dwordx4 here and the fragments came
                //  back wrong - NaN loss sums with correct gradients, round 6 - as they did from asm loads at the top of the block that
                //  nothing inside the tile loop named; scripts/lab_r06/lab_r06_nan.py, profiles/r06_xm_block_end_loads.txt)
                Rc[f][k] = *reinterpret_cast<const v4i_t __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(rc_lane[f] + k * 1024));
    }
    float lsumf[RF] = {0.f, 0.f}, csumf[RF] = {0.f, 0.f};
    const bool half_tiles = args.half_tiles != 0;
    float* red = reinterpret_cast<float*>(smem + NBUF * BUF);      // [8 fragment slots][4]: loss sum, cd sum, pair-set, image
    sfor<RF>([&](auto FI) {
        constexpr int f = FI.value;
        if (!act[f]) return;
        float* const dRj = dRp[f];
        float* base = dRj ? dRj + ((size_t)fn[f] * ntiles + ft[f]) * (32 * DP) + lane * 4 : nullptr;
        // x in the layout of dR (rows in registers, channel on the lane) = X * I (selector fragments), as in k_corr_main
        v4i_t sel[2];
#pragma unroll
        for (int sI = 0; sI < 2; ++sI) {
            f16x8 s8;
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] = (8 * h + j + 16 * sI == r) ? (_Float16)1.f : (_Float16)0.f;
            sel[sI] = __builtin_bit_cast(v4i_t, s8);
        }
        float lsum = 0.f, csum = 0.f;
        sfor<NDF>([&](auto DI) {
            constexpr int d = DI.value;
            float v[16];
            if constexpr (d < 2) {
                sfor<16>([&](auto I) { v[I.value] = agpr_read<ADR + f * 32 + d * 16 + I.value>(); });
            } else {
                const f32x16 dv = __builtin_bit_cast(f32x16, dRv[f]);
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = dv[i];
            }
            if (base && 32 * d + r < args.D) {
                if (half_tiles) {
                    // fp16 tiles (DgScatterSrc.half): elements 8s .. 8s+7 as one 16-byte piece, [2][64][8] per channel group - half the
                    // bytes here and in k_combine_out, and two store instructions per group instead of four
                    _Float16* hb = reinterpret_cast<_Float16*>(dRj) + ((size_t)fn[f] * ntiles + ft[f]) * (32 * DP) + d * 1024 + lane * 8;
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp) {
                        f16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[8 * sp + e];
                        // (ordinary stores: non-temporal ones cost k_combine_out, which reads these tiles 80 us later, 3 us - 34.0 -> 31.1 - and
                        //  save this kernel 1.3)
                        *reinterpret_cast<v4i_t*>(hb + sp * 512) = __builtin_bit_cast(v4i_t, o);
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(base + (d * 4 + g) * 256));
                    }
                }
            }
            // sum clamp(cd)(fd'' - shift) = sum_p <x_p, dR'_p>, sum cd = sum_p <x_p, sum_q y_q>   (dg_corr.hip "FOLD")
            const float cs = cs_pre[d];
            acc_t X8;
            mfma_h80(X8, Rc[f][2 * d], sel[0]);
            if constexpr (2 * d + 1 < NKC) mfma_h8(X8, Rc[f][2 * d + 1], sel[1]);
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(X8));
            const f32x16 X = __builtin_bit_cast(f32x16, X8);
#pragma unroll
            for (int i = 0; i < 16; ++i) { lsum = fmaf(X[i], v[i], lsum); csum = fmaf(X[i], cs, csum); }
        });
        lsumf[f] = lsum; csumf[f] = csum;
    });
    BL_T(3);
    // per fragment slot: its sums and whose they are; the first slot of every (pair-set, image) adds its run up, in slot order
#pragma unroll
    for (int f = 0; f < RF; ++f) {
        const float ls = wave_sum_dpp(lsumf[f]), cs = wave_sum_dpp(csumf[f]);
        if (lane == 0) {
            float* e = red + (wid * RF + f) * 4;
            e[0] = act[f] ? ls : 0.f; e[1] = act[f] ? cs : 0.f;
            reinterpret_cast<int*>(e)[2] = act[f] ? fj[f] : -1; reinterpret_cast<int*>(e)[3] = fn[f];
            reinterpret_cast<float**>(red + NW * RF * 4)[wid * RF + f] = partp[f];
        }
    }
    __syncthreads();
    BL_T(4);
    if (tid < NW * RF) {
        // every slot's record at once (independent reads: one LDS latency - walking the slots of the run one read after the other
        // was 1.3 us of every block, round-4 block log), then the run's sums in slot order, in registers
        // (floats read as floats, ints as ints: a 16-byte int-vector read of the records let hipcc's type-based alias analysis
        //  treat the float members as unrelated to the stores above - it stored ONE sum twice)
        const int* ri = reinterpret_cast<const int*>(red);
        float ea[NW * RF], eb[NW * RF];
        int ej[NW * RF], en[NW * RF];
#pragma unroll
        for (int s2 = 0; s2 < NW * RF; ++s2) { ea[s2] = red[s2 * 4]; eb[s2] = red[s2 * 4 + 1]; ej[s2] = ri[s2 * 4 + 2]; en[s2] = ri[s2 * 4 + 3]; }
        const int j = ri[tid * 4 + 2], nn = ri[tid * 4 + 3];
        const int pj = ri[(tid > 0 ? tid - 1 : 0) * 4 + 2], pn = ri[(tid > 0 ? tid - 1 : 0) * 4 + 3];
        float* const part_j = reinterpret_cast<float* const*>(red + NW * RF * 4)[tid];
        const bool first = j >= 0 && (tid == 0 || pj != j || pn != nn);
        if (first && part_j) {
            float a = 0.f, b = 0.f;
            bool in_run = true;
#pragma unroll
            for (int s2 = 0; s2 < NW * RF; ++s2) {
                const bool mine = s2 >= tid;
                in_run = in_run && (!mine || (ej[s2] == j && en[s2] == nn));
                if (mine && in_run) { a += ea[s2]; b += eb[s2]; }
            }
            float* part = part_j + (size_t)(nn * args.nrb + rb) * 2;
            part[0] = a; part[1] = b;
        }
    }
#ifdef C2_BLOCKLOG
    if (args.blocklog && threadIdx.x == 0) { unsigned long long* e = args.blocklog + (size_t)orig * 16; e[5] = wall_clock64(); for (int i = 0; i < 5; ++i) e[8 + i] = bl_t[i]; }
#endif
#ifdef C2_STAMPS
    BSTAMP(3);
    if (stamping) {
        __syncthreads();
        for (int i = tid; i < NW * 25 * 6 + NW * 4; i += 256) args.stamps[i] = st_lds[i];
    }
#endif
    advance();
    }       // (next work item)
    if (threadIdx.x == 0) dg_span_exit(args_k.span, span_keep);
    // the last workgroup to leave puts the counters back to zero (a re-launch on the same workspace - dg_corr_relaunch_main - finds
    // them as k_colmean left them for this one)
    if (DYN && threadIdx.x == 0) {
        uint32_t* const wctr = args_k.wctr;
        if (__hip_atomic_fetch_add(&wctr[8], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            for (int k = 0; k < 9; ++k) __hip_atomic_store(&wctr[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// blocks of a launch: per image the pair-sets' row blocks; with grouped ragged row blocks (args.gr_list)
// also the groups' blocks (the one-(pair-set, image) ragged blocks stay in the grid: they return at once when a group serves them)
static int dg_corr2_grid(const DgCorrArgs& args) {
    return args.njobs * args.B * args.nrb + (args.gr_list ? args.B * args.gr_blocks_per_image : 0);
}

// workgroups of the launch: one per CU (persistent: each walks the items blockIdx.x, + gridDim.x, ...), a multiple of 8 so that an
// item's XCD is its index modulo 8 at every step of the walk
static int dg_corr2_launch_grid(const DgCorrArgs& args) {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        ncu = n / 8 * 8;
#ifdef DG_DEVTOOLS
        if (const char* g = getenv("DG_C2_GRID")) ncu = atoi(g);     // developer A/B: 0 = one workgroup per item (the round-3 launch)
#endif
    }
    const int items = dg_corr2_grid(args);
    return (ncu <= 0 || items < ncu) ? items : ncu;
}

// which walk the persistent workgroups take: dynamic (per-XCD work counters) at six or more items per workgroup, the fixed round-robin
// below that.  DG_C2_WALK=dynamic|static overrides (read once; for tests/test_gpu_configs.py, which checks that both give the same bits)
static bool dg_corr2_dynamic_walk(const DgCorrArgs& args, int grid) {
    static const int forced = [] {
        const char* e = getenv("DG_C2_WALK");
        return !e ? 0 : (e[0] == 'd' ? 1 : (e[0] == 's' ? -1 : 0));
    }();
    const int items = dg_corr2_grid(args);
    if (args.wctr == nullptr || (grid & 7) != 0 || items <= grid) return false;
    if (forced) return forced > 0;
    return items >= 6 * grid;
}

// Helper jobs (stationary = operand 1) of a gradient pass with clamp(cd) = cd * mask.  Returns hipErrorNotSupported for
// shapes this form does not cover (the caller then uses k_corr_main).
// The shapes and clamp recipe this form covers - ONE predicate for the launcher below and for the host's plan (dg_api.hip decides
// from it whether the intra pair-set may be folded: a second copy of these conditions there could drift from this one and turn a
// fall-back to k_corr_main into a failed call)
bool dg_corr2_shape_supported(int KF, int KD, int D, float lo, float hi, int Ppad, int B) {
#ifdef C2_DISABLE          // (developer A/B: everything through k_corr_main)
    return false;
#endif
    return KF == 384 && KD == 96 && D <= 80 && lo == 0.f && hi > 1e30f && Ppad >= 160 && B <= 64;
}
bool dg_corr2_supported(const DgCorrArgs& args, int KF, int KD) {
    if (!dg_corr2_shape_supported(KF, KD, args.D, args.lo, args.hi, args.Ppad, args.B)) return false;
    for (int j = 0; j < args.njobs; ++j) {
        const DgJob& J = args.jobs[j];
        if (J.kind != DG_JOB_HELPER || !J.center_on_lane || !J.Gout || J.ridx) return false;
        if ((J.maskbits != nullptr) != (args.jobs[0].maskbits != nullptr)) return false;     // exact masks: for all pair-sets or none
    }
    return true;
}

hipError_t dg_launch_corr2(const DgCorrArgs& args, int KF, int KD, hipStream_t stream) {
    if (!dg_corr2_supported(args, KF, KD)) return hipErrorNotSupported;
    using BL = BlobT<24, 6>;
    if (args.njobs > 0 && args.jobs[0].maskbits) {
        // exact clamp masks (DG_EXACT_MASKS on the dense grid): the mask words of k_cd_mask instead of the sign of the fp16 cd;
        // 8 KiB more LDS for the four-slot word ring of the eight fragments
        const int smem_x = 4 * BL::BYTES + C2_RED_BYTES + 4 * 2 * 4 * 256;
        const int gx = dg_corr2_launch_grid(args);
        const bool dynx = dg_corr2_dynamic_walk(args, gx);
        bool foldx = false;
        for (int j = 0; j < args.njobs; ++j) foldx = foldx || args.jobs[j].fold != 0;
        auto kx = foldx ? (dynx ? k_corr2<24, 6, 5, true, true, true> : k_corr2<24, 6, 5, true, false, true>)
                        : (dynx ? k_corr2<24, 6, 5, true, true> : k_corr2<24, 6, 5, true, false>);
        hipError_t ex = dg_set_max_smem(reinterpret_cast<const void*>(kx), smem_x);
        if (ex != hipSuccess) return ex;
#if defined(DG_DEVTOOLS) && defined(C2_STAMPS)
        if (const char* stamp_file = getenv("DG_STAMPS")) {
            static uint32_t* stamp_buf = nullptr;
            if (!stamp_buf && hipMalloc(&stamp_buf, (4 * 25 * 6 + 16) * 4) != hipSuccess) return hipErrorOutOfMemory;
            DgCorrArgs a2 = args;
            a2.stamps = stamp_buf;
            (void)dg_set_max_smem(reinterpret_cast<const void*>(kx), smem_x + (4 * 25 * 6 + 16) * 4);
            hipLaunchKernelGGL(kx, dim3(gx), dim3(256), smem_x + (4 * 25 * 6 + 16) * 4, stream, a2);
            uint32_t host[4 * 25 * 6 + 16];
            if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(host, stamp_buf, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE* fp = fopen(stamp_file, "wb")) { fwrite(host, 4, 4 * 25 * 6 + 16, fp); fclose(fp); }
            return hipGetLastError();
        }
#endif
        hipLaunchKernelGGL(kx, dim3(gx), dim3(256), smem_x, stream, args);
        return hipGetLastError();
    }
    const int smem = 4 * BL::BYTES + C2_RED_BYTES;
    const int g0 = dg_corr2_launch_grid(args);
    const bool dyn0 = dg_corr2_dynamic_walk(args, g0);
    bool fold = false;
    for (int j = 0; j < args.njobs; ++j) fold = fold || args.jobs[j].fold != 0;
    auto kern = fold ? (dyn0 ? k_corr2<24, 6, 5, false, true, true> : k_corr2<24, 6, 5, false, false, true>)
                     : (dyn0 ? k_corr2<24, 6, 5, false, true> : k_corr2<24, 6, 5, false, false>);
    hipError_t e = dg_set_max_smem(reinterpret_cast<const void*>(kern), smem);
    if (e != hipSuccess) return e;
#if defined(DG_DEVTOOLS) && defined(C2_STAMPS)
    if (const char* stamp_file = getenv("DG_STAMPS")) {
        static uint32_t* stamp_buf = nullptr;
        if (!stamp_buf && hipMalloc(&stamp_buf, (4 * 25 * 6 + 16) * 4) != hipSuccess) return hipErrorOutOfMemory;
        DgCorrArgs a2 = args;
        a2.stamps = stamp_buf;
        (void)dg_set_max_smem(reinterpret_cast<const void*>(kern), smem + (4 * 25 * 6 + 16) * 4);
        hipLaunchKernelGGL(kern, dim3(dg_corr2_launch_grid(args)), dim3(256), smem + (4 * 25 * 6 + 16) * 4, stream, a2);
        uint32_t host[4 * 25 * 6 + 16];
        if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(host, stamp_buf, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* fp = fopen(stamp_file, "wb")) { fwrite(host, 4, 4 * 25 * 6 + 16, fp); fclose(fp); }
        return hipGetLastError();
    }
#endif
#if defined(DG_DEVTOOLS) && defined(C2_BLOCKLOG)
    if (const char* blog_file = getenv("DG_BLOCKLOG")) {
        static unsigned long long* blog_buf = nullptr;
        const int grid = dg_corr2_grid(args);
        if (!blog_buf && hipMalloc(&blog_buf, 8192 * 128) != hipSuccess) return hipErrorOutOfMemory;
        if (grid <= 8192) {
            DgCorrArgs a2 = args;
            a2.blocklog = blog_buf;
            (void)hipMemsetAsync(blog_buf, 0, (size_t)grid * 128, stream);
            hipLaunchKernelGGL(kern, dim3(dg_corr2_launch_grid(args)), dim3(256), smem, stream, a2);
            static unsigned long long hostb[8192 * 16];
            if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(hostb, blog_buf, (size_t)grid * 128, hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE* fp = fopen(blog_file, "wb")) { fwrite(hostb, 8, (size_t)grid * 16, fp); fclose(fp); }
            return hipGetLastError();
        }
    }
#endif
    hipLaunchKernelGGL(kern, dim3(dg_corr2_launch_grid(args)), dim3(256), smem, stream, args);
    return hipGetLastError();
}


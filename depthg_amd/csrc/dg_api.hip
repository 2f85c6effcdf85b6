// C ABI of the gfx950 DepthG correlation-loss library (declared in include/depthg_corr.h).
// Host-side orchestration only: workspace carving, job tables, kernel launches on the caller's stream.
#include "dg_common.h"

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <optional>

// ---- error reporting
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define DG_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(DG_ERR_LAUNCH, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int dg_version(void) { return DG_VERSION; }
extern "C" const char* dg_last_error(void) { return g_err; }

// ---- measurement aid: the fused correlation launch's execution span (include/depthg_corr.h dg_prof_main_span)
static unsigned long long* g_prof_span = nullptr;
extern "C" int dg_prof_main_span(void* span) { g_prof_span = static_cast<unsigned long long*>(span); return DG_OK; }

// ---- workspace plan
static inline size_t up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Plan {
    int B, C, D, h, w, hc, wc, S, Sh, P, Ppad, KF, KD, C4, D4, N, T, nops, rf, nrb, blob;     // (hc, wc): size of the code maps
    bool shared, depth, grad, pointwise, ident, rows;
    size_t nhwc_f[2], nhwc_c[2];
    size_t rows_f[DG_MAX_NEG + 2], rows_c[DG_MAX_NEG + 2];     // sampled fp32 rows per operand (small sample grids)
    size_t op[DG_MAX_NEG + 2], inv[DG_MAX_NEG + 2], colpart[DG_MAX_NEG + 2], bbar[DG_MAX_NEG + 2];
    size_t ccolpart[DG_MAX_NEG + 2], csum[DG_MAX_NEG + 2], bsplit[DG_MAX_NEG + 2];
    size_t rvec[DG_MAX_NEG + 2], rimg[DG_MAX_NEG + 2];
    size_t nz, nzsum;
    size_t dRA[DG_MAX_NEG + 3], dRB[DG_MAX_NEG + 2];   // dRA[T] = depth job
    size_t part[DG_MAX_NEG + 3];
    size_t comb[2], scratch_out, taps, gbuf[DG_MAX_NEG + 2];
    size_t ticket;                          // the depth blocks' ticket of the k_gs launch
    size_t maskbits[DG_MAX_NEG + 2];        // exact clamp masks of the pair-sets (k_cd_mask), xmask: in use on a small sample grid,
    bool half;                              // fp16 gradient tiles between k_corr2 / k_gs and k_combine_out (identity grid; DgScatterSrc.half)
    bool xmask, xmask_dense;                // xmask_dense: on the dense identity grid (DG_EXACT_MASKS)
    bool fold;                              // the intra pair-set's streamed-side gradient is formed in the fused kernel (dg_corr2.hip FOLD)
    size_t clo[2];                          // ... with pointwise: the parts of the normalised code the fp16 C parts drop (k_cd_mask3)
    size_t gr_list, gr_count, gr_rank;      // consumer lists of the grouped ragged row blocks (dg_corr2.hip)
    // the fused small-grid path (dg_small.hip): sampled rows -> ONE launch
    bool small;
    int nsplit;                             // blocks per (image, pair-set): 2 when the image has 5 tiles
    size_t dRA2[DG_MAX_NEG + 2], dRBs[DG_MAX_NEG + 2], dRB2[DG_MAX_NEG + 2][2], dRBm[DG_MAX_NEG + 2], part4, om;
    size_t total;
};

// developer A/B: DG_SMALL_PATH=0 keeps the multi-launch small-grid path of rounds 2-4 (needs C <= 768)
#define FOLD_STASH_OFF (5 * 1024)        // k_corr2<24, 6, 5>: code k-step 5 of the C part (channels 80 .. 95: padding for D <= 80)
static bool small_path_enabled() {
    static const bool on = [] { const char* e = getenv("DG_SMALL_PATH"); return !(e && e[0] == '0'); }();
    return on;
}

static void clamp_bounds(const dg_corr_desc* d, float& lo, float& hi);
static int make_plan(const dg_corr_desc* d, Plan& p) {
    if (!d) return fail(DG_ERR_INVALID, "null descriptor");
    if (d->B < 1 || d->C < 1 || d->D < 1 || d->h < 1 || d->w < 1 || d->S < 1)
        return fail(DG_ERR_INVALID, "non-positive dimension in descriptor");
    if (d->n_neg < 0 || d->n_neg > DG_MAX_NEG) return fail(DG_ERR_UNSUPPORTED, "n_neg=%d outside [0,%d]", d->n_neg, DG_MAX_NEG);
    if (d->C > 8192) return fail(DG_ERR_UNSUPPORTED, "C=%d > 8192 feature channels not supported", d->C);
    if (d->D > 128) return fail(DG_ERR_UNSUPPORTED, "D=%d > 128 code channels not supported", d->D);
    if ((size_t)d->h * d->w > 16384) return fail(DG_ERR_UNSUPPORTED, "feature map %dx%d too large", d->h, d->w);
    if (d->code_h < 0 || d->code_w < 0 || (d->code_h == 0) != (d->code_w == 0))
        return fail(DG_ERR_INVALID, "code_h=%d, code_w=%d: both zero (code maps of the feature maps' size) or both positive", d->code_h, d->code_w);
    p.B = d->B; p.C = d->C; p.D = d->D; p.h = d->h; p.w = d->w; p.S = d->S; p.N = d->n_neg;
    p.hc = d->code_h ? d->code_h : d->h; p.wc = d->code_w ? d->code_w : d->w;
    if ((size_t)p.hc * p.wc > 16384) return fail(DG_ERR_UNSUPPORTED, "code map %dx%d too large", p.hc, p.wc);
    const bool same_maps = p.hc == p.h && p.wc == p.w;
    p.Sh = (d->flags & DG_LINE_GRID) ? 1 : d->S;      // sample grid: Sh rows x S columns
    p.P = p.Sh * d->S;
    p.Ppad = (int)up(p.P, 32);
    p.KF = d->C <= 128 ? 128 : (d->C <= 384 ? 384 : 768);
    p.KD = d->D <= 96 ? 96 : 128;
    // Sample grids of at most 160 positions (every recipe the reference ships: feature_samples = 11 / 12) take the fused small-grid
    // kernel, which streams the feature channels in chunks and so has no width limit (FeaturePyramidNet: 2048).  The blob kernels
    // (larger grids, the identity grid) hold whole channel vectors in registers / LDS: C <= 768 there.
    p.small = !(d->flags & DG_IDENTITY_GRID) && dg_small_supported(p.Ppad, p.KD) && small_path_enabled();
    p.nsplit = p.Ppad == 160 ? 2 : 1;
    if (d->C > 768 && !p.small)
        return fail(DG_ERR_UNSUPPORTED, "C=%d > 768 feature channels are supported on sample grids of at most 160 positions "
                                        "(feature_samples <= 12) only; this call has %d positions%s", d->C, p.P,
                    (d->flags & DG_IDENTITY_GRID) ? " on the dense identity grid (wider maps go there in channel chunks: dg_normalize_split + DG_FEATS_UNIT)"
                                                   : " (wider maps go there in channel chunks: dg_sampled_sumsq + dg_corr_forward_extnorm)");
    if ((d->flags & DG_FEATS_UNIT) && !(d->flags & DG_IDENTITY_GRID))
        return fail(DG_ERR_INVALID, "DG_FEATS_UNIT needs DG_IDENTITY_GRID: on sampled coordinates the reference normalises BEHIND sample()");
    p.C4 = (int)up(d->C, 4); p.D4 = (int)up(d->D, 4);
    p.T = 2 + p.N;
    p.shared = (d->flags & DG_SHARED_COORDS) != 0;
    p.depth = (d->flags & DG_DEPTH_TERM) != 0;
    p.grad = (d->flags & DG_NEED_GRAD) != 0;
    p.pointwise = (d->flags & DG_POINTWISE) != 0;
    p.ident = (d->flags & DG_IDENTITY_GRID) != 0;
    if (p.ident && (p.Sh != p.S || !p.shared || d->S != d->h || d->S != d->w || d->w > 64))
        return fail(DG_ERR_INVALID, "DG_IDENTITY_GRID needs DG_SHARED_COORDS and S == h == w <= 64");
    if (p.ident && !same_maps)
        return fail(DG_ERR_INVALID, "DG_IDENTITY_GRID needs code maps of the feature maps' size (got %dx%d against %dx%d)", p.hc, p.wc, p.h, p.w);
    p.nops = p.shared ? 2 : p.T;
    p.rf = (p.KF == 384 && p.KD == 96 && p.Ppad > 128) ? 8 : 4;    // waves per block (32 stationary rows each)
    p.nrb = (p.Ppad + p.rf * 32 - 1) / (p.rf * 32);
    p.blob = DgBlob(p.KF, p.KD).bytes;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += up(bytes, 256); return o; };
    const size_t HW = (size_t)p.h * p.w, HWc = (size_t)p.hc * p.wc, B = p.B;
    // small sample grids (all operands together sample fewer positions than the two maps have pixels; planes of 32 channels
    // fit the LDS; batch indices fit the 16-bit consumer lists): sample() straight from NCHW (k_plane_sample) instead of
    // channel-last copies of the whole maps
    // (code maps of another size than the feature maps - the FeaturePyramidNet contract - take the channel-last gather path)
    p.rows = !p.ident && same_maps && (size_t)p.nops * p.P <= 2 * HW && HW <= 1024 && B <= 32767;
    for (int i = 0; i < 2; ++i) { p.nhwc_f[i] = take(p.rows ? 0 : B * HW * p.C4 * 4); p.nhwc_c[i] = take(p.rows ? 0 : B * HWc * p.D4 * 4); }
    const bool want_rows = p.rows || p.small;      // (the fused small-grid kernel reads sampled rows whichever sampler wrote them)
    // (the fused small-grid kernel takes bf16 feature rows of up(C, 128) channels - whole chunks; the multi-launch path fp32 rows of C4)
    const size_t frow = p.small ? (size_t)up(p.C, 128) * 2 : (size_t)p.C4 * 4;
    for (int i = 0; i < p.nops; ++i) { p.rows_f[i] = take(want_rows ? B * p.P * frow : 0); p.rows_c[i] = take(want_rows ? B * p.P * p.D4 * 4 : 0); }
    for (int i = 0; i < p.nops; ++i) {
        p.op[i] = take(B * (p.Ppad / 32) * (size_t)p.blob);
        p.inv[i] = take(B * p.Ppad * 4);
        p.colpart[i] = take(B * (size_t)(p.ident ? p.h * ((p.w + 31) / 32) : p.Ppad / 32) * p.KF * 4);
        p.bbar[i] = take(B * p.KF * 4);
        p.bsplit[i] = take(B * 2 * p.KF * 2);
        p.ccolpart[i] = take(B * (size_t)(p.Ppad / 32) * p.KD * 4);
        p.csum[i] = take(B * p.KD * 4);
    }
    for (int t = 0; t < p.T; ++t) { p.rvec[t] = take(B * p.Ppad * 4); p.rimg[t] = take(B * 4); }
    p.nz = take(B * p.Ppad * 4);
    p.nzsum = take(B * 4);
    for (int t = 0; t <= p.T; ++t) { p.dRA[t] = take(B * p.Ppad * p.KD * 4); p.part[t] = take(B * p.nrb * 2 * 4); }
    for (int t = 0; t < p.T; ++t) p.dRB[t] = take(B * p.Ppad * p.KD * 4);
    for (int i = 0; i < 2; ++i) p.comb[i] = take(B * p.Ppad * p.KD * 4);
    p.scratch_out = take(DG_OUT_COUNT * 4);
    p.taps = take(2 * B * dg_taps_record_bytes(p.hc * p.wc, p.P));     // the adjoint of sample() scatters into the CODE maps
    for (int t = 0; t < p.T; ++t) p.gbuf[t] = p.grad ? take(B * (size_t)(p.Ppad / 32) * (p.Ppad / 32) * 2048) : 0;     // (gbuf[0]: 4 KiB would do with p.fold, which is decided below)
    p.ticket = take(256);
    // exact clamp masks: gradient passes of the zero_clamp recipe on small sample grids (fp32 sampled rows exist, <= 8 tiles, the
    // one-wave-per-SIMD form of k_corr_main)
    p.xmask = p.rows && !p.small && p.grad && (d->flags & DG_ZERO_CLAMP) && !(d->flags & DG_STABALIZE) && p.Ppad <= 256 && p.rf == 4;
#ifdef DG_NO_XMASK      // developer A/B: the fp16 masks everywhere
    p.xmask = false;
#endif
    // DG_EXACT_MASKS: the dense identity grid at the widths of the one-wave-per-SIMD kernel (dg_corr2.hip) takes the same mask words,
    // computed from channel-last fp32 copies of the two code maps (the workspace's nhwc_c regions, otherwise unused on this grid)
    p.xmask_dense = (d->flags & DG_EXACT_MASKS) && p.ident && p.grad && (d->flags & DG_ZERO_CLAMP) && !(d->flags & DG_STABALIZE) &&
                    p.KF == 384 && p.KD == 96 && p.D <= 80 && p.Ppad >= 160 && p.B <= 64;
    if ((d->flags & DG_EXACT_MASKS) && p.grad && (d->flags & DG_ZERO_CLAMP) && !p.xmask && !p.xmask_dense && !p.small)
        return fail(DG_ERR_UNSUPPORTED, "DG_EXACT_MASKS: exact clamp masks exist on small sample grids (always on there) and on the dense "
                                        "identity grid with C <= 384 (padded to 384), D <= 80, P >= 160, B <= 64, zero_clamp without stabalize");
    for (int t = 0; t < p.T; ++t) p.maskbits[t] = take((p.xmask || p.xmask_dense) ? B * (size_t)(p.Ppad / 32) * p.Ppad * 4 : 0);
    // FOLD: gradient passes of the pointwise recipe that k_corr2 runs (dg_corr2_shape_supported: the launcher's own predicate; the
    // job-level conditions - stationary = operand 1, G tiles wanted, no batch map on R - hold for every gradient pass); DG_FOLD_INTRA=0 keeps the k_gs job (developer A/B)
    {
        static const bool fold_on = [] { const char* e = getenv("DG_FOLD_INTRA"); return !(e && e[0] == '0'); }();
        float lo, hi;
        clamp_bounds(d, lo, hi);
        p.fold = fold_on && p.grad && p.pointwise && !p.small && dg_corr2_shape_supported(p.KF, p.KD, p.D, lo, hi, p.Ppad, p.B) &&
                 !p.xmask;          // (with the dense grid's exact mask words too, since round 6: k_corr2<.., XM, .., FOLD>)
    }
    // fp16 gradient tiles (round 6): the identity grid's backward is ONE launch (k_combine_out) that reads the raw tiles of k_corr2 and the
    // streamed-side tiles of k_gs once - 93 of the headline step's 1342 MB go with fp32 -> fp16 (both producers bounded: the raw tiles by
    // construction, k_gs's by leaving the division by ||c|| to the consumer).  Where k_corr2 runs and k_combine_out will (its routed list
    // holds n_neg x B entries at most 512); DG_HALF_TILES=0 keeps fp32 tiles (developer A/B)
    {
        static const bool half_on = [] { const char* e = getenv("DG_HALF_TILES"); return !(e && e[0] == '0'); }();
        float lo, hi;
        clamp_bounds(d, lo, hi);
        p.half = half_on && p.ident && p.grad && !p.small && dg_corr2_shape_supported(p.KF, p.KD, p.D, lo, hi, p.Ppad, p.B) &&
                 p.N * p.B <= 512 && p.S == p.h && p.S == p.w;
    }
    for (int i = 0; i < 2; ++i) p.clo[i] = take((p.xmask_dense && p.pointwise) ? B * (size_t)(p.Ppad / 32) * p.KD * 64 : 0);
    p.gr_list = take((size_t)DG_MAX_JOBS * B * DG_GR_CAP * 4);
    p.gr_count = take((size_t)DG_MAX_JOBS * B * 4);
    p.gr_rank = take((size_t)DG_MAX_JOBS * B * 2);
    {
        const bool two = p.small && p.grad && p.pointwise;       // the old_mean terms of the gradient (dg_small.hip)
        const size_t gt = B * p.Ppad * p.KD * 4;
        for (int t = 0; t < p.T; ++t) {
            p.dRA2[t] = take(two ? gt : 0);
            p.dRBs[t] = take(p.small && p.grad && p.nsplit == 2 ? gt : 0);
            for (int k = 0; k < 2; ++k) p.dRB2[t][k] = take(two && k < p.nsplit ? gt : 0);
            p.dRBm[t] = take(p.small && p.grad && t >= 2 && (p.pointwise || p.nsplit == 2) ? gt : 0);
        }
        p.part4 = take(p.small ? (size_t)(p.T + 1) * B * p.nsplit * 16 : 0);
        p.om = take(p.small ? (size_t)(p.T + 1) * 4 : 0);
    }
    p.total = off;
    return DG_OK;
}

extern "C" size_t dg_corr_workspace_bytes(const dg_corr_desc* desc) {
    Plan p;
    if (make_plan(desc, p) != DG_OK) return 0;
    return p.total;
}

// operand index used as the S operand (pass A) of pair-set t, and its batch map
static inline int op_of(const Plan& p, int t) { return t < 2 ? t : (p.shared ? 0 : t); }
static inline const int64_t* map_of(const Plan& p, int t, const int64_t* perms) {
    return (t >= 2 && p.shared) ? perms + (size_t)(t - 2) * p.B : nullptr;
}

static void clamp_bounds(const dg_corr_desc* d, float& lo, float& hi) {
    lo = (d->flags & DG_ZERO_CLAMP) ? 0.0f : -9999.0f;
    hi = (d->flags & DG_STABALIZE) ? 0.8f : __builtin_inff();
}

static float shift_of(const dg_corr_desc* d, int t) { return t == 0 ? d->shift_intra : (t == 1 ? d->shift_inter : d->shift_neg); }

// Fills one helper job.  passB: the stationary operand is operand 2 of pair-set t.
static DgJob helper_job(const Plan& p, const dg_corr_desc* d, char* ws, int t, bool passB, const int64_t* perms) {
    DgJob j;
    memset(&j, 0, sizeof(j));
    const int o2 = op_of(p, t);
    const int64_t* m2 = map_of(p, t, perms);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    if (!passB) {
        j.Rop = ws + p.op[0]; j.RcInv = F32(p.inv[0]); j.ridx = nullptr;
        j.Sop = ws + p.op[o2]; j.sidx = m2;
        j.Scsum = F32(p.csum[o2]);
        j.center_on_lane = 1;
    } else {
        j.Rop = ws + p.op[o2]; j.RcInv = F32(p.inv[o2]); j.ridx = m2;
        j.Sop = ws + p.op[0]; j.sidx = nullptr;
        j.center_on_lane = 0;
    }
    if (p.pointwise) { j.rvec = F32(p.rvec[t]); j.rimg = F32(p.rimg[t]); }
    j.shift = shift_of(d, t);
    j.kind = DG_JOB_HELPER;
    return j;
}

static DgJob depth_job(const Plan& p, const dg_corr_desc* d, char* ws) {
    DgJob j;
    memset(&j, 0, sizeof(j));
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    j.Rop = ws + p.op[0]; j.Sop = ws + p.op[0]; j.RcInv = F32(p.inv[0]);
    j.nzR = F32(p.nz); j.nzS = F32(p.nz);
    j.shift = d->shift_depth;
    j.kind = DG_JOB_DEPTH;
    j.center_on_lane = 1;
    return j;
}

static void corr_args_base(const Plan& p, const dg_corr_desc* d, char* ws, DgCorrArgs& a) {
    memset(&a, 0, sizeof(a));
    a.B = p.B; a.P = p.P; a.Ppad = p.Ppad; a.nrb = p.nrb; a.D = p.D;
    clamp_bounds(d, a.lo, a.hi);
    a.inv_BP = 1.0f / ((float)p.B * (float)p.P);
    a.dummy = ws + p.op[0];
    a.span = g_prof_span;
    a.half_tiles = p.half ? 1 : 0;
}

// k_gs jobs: one per pair-set; the producing job of the G tiles is helper_job(t) of the fused launch (R = operand 1,
// S = operand 2 of pair-set t)
// A second stream of the library's own (one per device, created on the first call that is not being captured into a graph) for the
// launches that may run beside each other inside one call; null while none exists and the caller's stream is capturing (creating
// one there is not a capturable operation: the call then launches in sequence).
struct SideStream { hipStream_t s; hipEvent_t fork, join, mid[2]; std::mutex use; };
static SideStream* side_stream_for(hipStream_t caller) {
    static std::mutex mu;
    static std::map<int, SideStream> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
#ifdef DG_DEVTOOLS
    if (const char* e = getenv("DG_SIDE_STREAM")) if (e[0] == '0') return nullptr;       // (developer A/B: every call launches in sequence)
#endif
    std::lock_guard<std::mutex> lk(mu);
    auto it = table.find(dev);
    if (it != table.end()) return &it->second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(caller, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    hipStream_t s = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};       // fork, join, two hand-overs from the side stream in mid-region
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    bool ok = true;
    for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        for (int i = 0; i < 4; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);      // nothing half-made stays behind: the call launches in sequence instead
        (void)hipStreamDestroy(s);
        return nullptr;
    }
    SideStream& ss = table[dev];        // (std::map nodes do not move: the pointer stays valid; std::mutex is not copyable)
    ss.s = s; ss.fork = ev[0]; ss.join = ev[1]; ss.mid[0] = ev[2]; ss.mid[1] = ev[3];
    return &ss;
}

// One fork .. join region on the device's side stream.  The stream and its event pair are shared by every caller on the device, so
// the region holds the stream's lock from the fork record to the join wait: two host threads (or two caller streams) cannot interleave
// their records and waits.  Whatever happens after the fork - a failed launch returns through DG_HIP - the destructor still records
// the join and makes the caller's stream wait for it: the side stream is never left unjoined (inside a hipGraph capture that would be
// a forked capture that cannot end).
struct SideRegion {
    SideStream* side;
    hipStream_t caller;
    std::unique_lock<std::mutex> lk;
    bool forked = false, join_recorded = false, joined = false;
    explicit SideRegion(hipStream_t caller_) : side(side_stream_for(caller_)), caller(caller_) {
        if (side) lk = std::unique_lock<std::mutex>(side->use);
    }
    explicit operator bool() const { return side != nullptr; }
    hipStream_t stream() const { return side->s; }
    hipError_t fork() {
        hipError_t e = hipEventRecord(side->fork, caller);
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(side->s, side->fork, 0);
        forked = e == hipSuccess;
        return e;
    }
    void reset() { forked = join_recorded = joined = false; }         // (after a join: the region may fork again)
    hipError_t record_join() {
        hipError_t e = hipEventRecord(side->join, side->s);
        join_recorded = e == hipSuccess;
        return e;
    }
    // hand-over i in mid-region: everything launched on the side stream so far is ordered in front of what the caller launches next
    hipError_t hand_over(int i) {
        hipError_t e = hipEventRecord(side->mid[i], side->s);
        if (e != hipSuccess) return e;
        return hipStreamWaitEvent(caller, side->mid[i], 0);
    }
    hipError_t join() {
        if (!forked || joined) return hipSuccess;
        if (!join_recorded) { hipError_t e = record_join(); if (e != hipSuccess) return e; }
        hipError_t e = hipStreamWaitEvent(caller, side->join, 0);
        joined = e == hipSuccess;
        return e;
    }
    ~SideRegion() { if (side && forked && !joined) (void)join(); }
};

// Without `pointwise` the intra pair-set (t = 0) has NO job here (round 4): it correlates the anchors with themselves at the same
// coordinates, so fd, cd and with them -G are symmetric and the gradient through the streamed side equals the one through the
// stationary side, which the fused kernel accumulates in registers anyway - the backward doubles that one (as it always did for the
// depth term) instead of reading 1/7 of the G tiles again.  With `pointwise` -G is NOT symmetric: the reference centres fd by its
// ROW means only (fd -= fd.mean([3, 4]), src/modules.py:1238-1239), -G[p][q] - -G[q][p] = mask (rowmean_q - rowmean_p) - invisible
// on i.i.d. features, 1e-2 of the gradient on the FPS recipes (the test with exact masks caught it).
// (p.fold: with `pointwise` k_corr2 forms G + G^T in its own accumulator, dg_corr2.hip FOLD - the same consequence for the launches)
static bool intra_is_symmetric(const Plan& p) { return !p.pointwise || p.fold; }
static void build_gs_jobs(const Plan& p, char* ws, const int64_t* perms, DgGsArgs& g) {
    memset(&g, 0, sizeof(g));
    const int t0 = intra_is_symmetric(p) ? 1 : 0;
    g.njobs = p.T - t0; g.B = p.B; g.P = p.P; g.Ppad = p.Ppad; g.KF = p.KF; g.KD = p.KD; g.D = p.D;
    for (int t = t0; t < p.T; ++t) {
        const int o2 = op_of(p, t);
        DgGsJob& J = g.jobs[t - t0];
        J.G = reinterpret_cast<const uint16_t*>(ws + p.gbuf[t]);
        J.Rop = ws + p.op[0]; J.ridx = nullptr;
        J.Sop = ws + p.op[o2]; J.sidx = map_of(p, t, perms);
        J.ScInv = reinterpret_cast<const float*>(ws + p.inv[o2]);
        J.dS = reinterpret_cast<float*>(ws + p.dRB[t]);
    }
}

// Job table of the fused correlation launch: one job per pair-set (stationary = operand 1); on forward-only calls the cheap
// depth job last.
// Returns the number of pair-set jobs; *depth_index = position of the depth job or -1.  (Stationary = operand 2 is only
// used by dg_corr_materialize, whose stores then run along the second position index.)
static int build_corr_jobs(const Plan& p, const dg_corr_desc* desc, char* ws, const int64_t* perms, DgCorrArgs& a,
                           int* depth_index) {
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    corr_args_base(p, desc, ws, a);
    a.wctr = p.grad ? reinterpret_cast<uint32_t*>(ws + p.ticket) + 16 : nullptr;      // (behind the depth blocks' ticket word)
    const double numel = (double)p.B * p.P * p.P;
    int nj = 0;
    for (int t = 0; t < p.T; ++t) {
        DgJob j = helper_job(p, desc, ws, t, false, perms);
        j.part = F32(p.part[t]);
        j.dR = p.grad ? F32(p.dRA[t]) : nullptr;
        j.Gout = p.grad ? reinterpret_cast<uint16_t*>(ws + p.gbuf[t]) : nullptr;
        j.fold = (p.fold && t == 0) ? 1 : 0;
        j.maskbits = (p.xmask || p.xmask_dense) ? reinterpret_cast<const uint32_t*>(ws + p.maskbits[t]) : nullptr;
        j.slot_loss = t < 2 ? t : DG_OUT_LOSS_NEG;
        j.slot_cd = t < 2 ? DG_OUT_CD_INTRA + t : DG_OUT_CD_NEG;
        j.fin_scale = (float)(1.0 / (t < 2 ? numel : numel * p.N));
        a.jobs[nj++] = j;
    }
    const int njA = nj;
    // (the gradient of the streamed operand's code comes from k_gs, which consumes the G tiles these jobs store)
    *depth_index = -1;
    // (on a gradient pass the depth term runs as blocks of the k_gs launch, dg_corr.hip gs_depth_block: in the fused kernel's
    //  launch its latency-bound blocks were the tail)
    if (p.depth && !p.grad) {
        DgJob j = depth_job(p, desc, ws);
        j.part = F32(p.part[p.T]);
        j.dR = nullptr;
        j.slot_loss = DG_OUT_LOSS_DEPTH; j.slot_cd = -1; j.fin_scale = (float)(1.0 / numel);
        *depth_index = nj;
        a.jobs[nj++] = j;
    }
    a.njobs = nj;
    // Ragged last row blocks grouped by streamed operand (dg_corr2.hip): pair-sets that stream the same operand array form a
    // key.  Worth it when the ragged row block is short (at most 4 of the 8 row tiles) and several pair-sets share an array
    // (shared coordinates: intra + the negatives stream operand 0 through batch maps).
    a.gr_list = nullptr;
#ifndef DG_NO_GROUP          // (developer A/B: scripts/build_variant.sh with SRC=dg_api)
    {
        const int nt = p.Ppad / 32, L = nt % 8;
        if (p.grad && p.KF == 384 && p.KD == 96 && p.D <= 80 && p.nrb > 1 && L >= 1 && L <= 4 && p.B % 8 == 0 && p.B <= 64 && njA >= 2) {
            int nkeys = 0, members[DG_MAX_JOBS] = {0};
            for (int j = 0; j < njA; ++j) {
                int k = -1;
                for (int q = 0; q < nkeys; ++q) if (a.jobs[(int)a.gr_first[q]].Sop == a.jobs[j].Sop) k = q;
                if (k < 0) { k = nkeys++; a.gr_first[k] = (int8_t)j; }
                a.gr_key[j] = (int8_t)k; ++members[k];
            }
            bool shared_any = false;
            for (int k = 0; k < nkeys; ++k) shared_any |= members[k] > 1;
            if (shared_any) {
                a.gr_nkeys = nkeys; a.gr_cpb = 8 / L; a.gr_blocks_per_image = 0;
                for (int k = 0; k < nkeys; ++k) {
                    const bool single = members[k] == 1 && a.jobs[(int)a.gr_first[k]].sidx == nullptr;      // exactly one consumer per image
                    int nb = single ? 1 : (5 * members[k] / 2 + a.gr_cpb - 1) / a.gr_cpb;           // 2.5 x the mean consumer count
                    const int cap = (DG_GR_CAP + a.gr_cpb - 1) / a.gr_cpb;
                    a.gr_nblk[k] = nb > cap ? cap : nb;
                    a.gr_blocks_per_image += a.gr_nblk[k];
                }
                a.gr_list = reinterpret_cast<const int32_t*>(ws + p.gr_list);
                a.gr_count = reinterpret_cast<const int32_t*>(ws + p.gr_count);
                a.gr_rank = reinterpret_cast<const int16_t*>(ws + p.gr_rank);
            }
        }
    }
#endif
    return njA;
}

// Fused correlation launch.  Gradient passes of the ViT-S widths run the one-wave-per-SIMD kernel (dg_corr2.hip); everything else
// (forward-only calls, stabalize / no zero_clamp, ViT-B widths, small P) k_corr_main.
static hipError_t launch_main(const Plan& p, const DgCorrArgs& a, int njA, int depth_index, hipStream_t stream) {
    (void)depth_index;
    if (p.grad && njA > 0) {
        const hipError_t e = dg_launch_corr2(a, p.KF, p.KD, stream);      // the pair-set jobs, one launch
        if (e != hipErrorNotSupported) return e;
        if (p.fold || p.half) return hipErrorInvalidValue;          // (the plan folded the intra pair-set / asked for fp16 tiles of a kernel that does not run)
    }
    return dg_launch_corr(a, p.KF, p.KD, p.rf, p.grad ? 1 : 0, stream);
}

struct DrawArgs { int64_t* out; uint64_t seed; unsigned long long* state; };

// (DG_SPLIT_MASKS=0: the exact-mask chain of the dense grid in sequence on the caller's stream, developer A/B)
static bool split_masks_enabled() {
    static const bool on = [] { const char* e = getenv("DG_SPLIT_MASKS"); return !(e && e[0] == '0'); }();
    return on;
}

// ---- the fused small-grid path (dg_small.hip)
static void small_args(const Plan& p, const dg_corr_desc* d, char* ws, const int64_t* perms, DgSmallArgs& a) {
    memset(&a, 0, sizeof(a));
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    for (int o = 0; o < p.nops; ++o) { a.rowsF[o] = F32(p.rows_f[o]); a.rowsC[o] = F32(p.rows_c[o]); }
    a.T = p.T; a.B = p.B; a.P = p.P; a.Ppad = p.Ppad; a.C4 = (int)up(p.C, 128); a.D = p.D; a.D4 = p.D4; a.KD = p.KD;
    a.pointwise = p.pointwise ? 1 : 0; a.depth = p.depth ? 1 : 0; a.grad = p.grad ? 1 : 0;
    clamp_bounds(d, a.lo, a.hi);
    for (int t = 0; t < p.T; ++t) { a.shift[t] = shift_of(d, t); a.opS[t] = op_of(p, t); a.sidx[t] = map_of(p, t, perms); }
    a.shift_depth = d->shift_depth;
    a.nz = F32(p.nz); a.nzsum = F32(p.nzsum);
    for (int t = 0; t <= p.T; ++t) a.dRA[t] = F32(p.dRA[t]);
    for (int t = 0; t < p.T; ++t) {
        a.dRA2[t] = F32(p.dRA2[t]);
        a.dRB[t][0] = F32(p.dRB[t]); a.dRB[t][1] = F32(p.dRBs[t]);
        a.dRB2[t][0] = F32(p.dRB2[t][0]); a.dRB2[t][1] = F32(p.dRB2[t][1]);
    }
    a.part = F32(p.part4); a.om = F32(p.om);
    a.ticket = reinterpret_cast<unsigned int*>(ws + p.ticket) + 32;
    const DgBlob bl(p.KF, p.KD);
    a.xop = ws + p.op[0]; a.xinv = F32(p.inv[0]); a.blob_bytes = bl.bytes; a.blob_off_c = bl.off_c;
    a.wtot[0] = d->w_intra; a.wtot[1] = d->w_inter; a.wtot[2] = d->w_neg; a.wtot[3] = d->w_depth;
    a.nsplit = p.nsplit;
    a.span = g_prof_span;
#ifdef DG_DEVTOOLS
    { static const int dbg = [] { const char* e = getenv("DG_SMALL_DEBUG"); return e ? atoi(e) : 0; }(); a.debug = dbg; }
#endif
}

// sampled rows of every operand (the reference's sample(), src/modules.py:822-825, of feats / code at coords1 / coords2 and of the
// negatives' permuted maps, :1323-1343) -> the fused kernel.  Launches: the draw + depth indicators + tap records, the sampler, the
// fused kernel, its one-wave finish.
static int forward_small(const Plan& p, const dg_corr_desc* desc, const float* orig_feats, const float* orig_feats_pos,
                         const float* orig_code, const float* orig_code_pos, const float* depth, const float* coords1,
                         const float* coords2, const int64_t* perms, const DrawArgs* draw, float* out_scalars, char* ws,
                         hipStream_t stream) {
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    DgSmallArgs a;
    small_args(p, desc, ws, perms, a);
    a.out = out_scalars;
    {
        DgPreArgs q;
        memset(&q, 0, sizeof(q));
        if (draw && p.N > 0) { q.seed = draw->seed; q.state = draw->state; q.perms = draw->out; q.count = p.N; }
        if (p.depth) { q.depth = depth; q.nz = F32(p.nz); q.nzsum = F32(p.nzsum); q.dH = desc->depth_h; q.dW = desc->depth_w; }
        if (p.grad && (size_t)p.hc * p.wc <= 4096 && p.P <= 65535) { q.coords1 = coords1; q.coords2 = coords2; q.taps = ws + p.taps; }
        q.B = p.B; q.h = p.hc; q.w = p.wc; q.S = p.S; q.Sh = p.Sh; q.P = p.P; q.Ppad = p.Ppad;
        if (p.B > 8192 && q.count > 0) return fail(DG_ERR_UNSUPPORTED, "B=%d too large for the in-call draw", p.B);
        DG_HIP(dg_launch_pre_general(q, stream));
    }
    if (p.rows) {
        DgPlaneArgs t;
        memset(&t, 0, sizeof(t));
        t.src[0] = orig_feats; t.src[1] = orig_feats_pos; t.src[2] = orig_code; t.src[3] = orig_code_pos;
        t.K[0] = t.K[1] = p.C; t.K4[0] = t.K4[1] = (int)up(p.C, 128); t.K[2] = t.K[3] = p.D; t.K4[2] = t.K4[3] = p.D4;
        t.feats_bf16 = 1;
        for (int o = 0; o < p.nops; ++o) { t.rows[o][0] = F32(p.rows_f[o]); t.rows[o][1] = F32(p.rows_c[o]); }
        t.coords1 = coords1; t.coords2 = coords2; t.perms = perms;
        t.nops = p.nops; t.B = p.B; t.h = p.h; t.w = p.w; t.S = p.S; t.Sh = p.Sh; t.P = p.P;
        DG_HIP(dg_launch_plane_sample(t, stream));
    } else {
        // code maps of another size than the feature maps (the FeaturePyramidNet contract) or maps beyond the plane sampler's LDS:
        // channel-last copies, then a bilinear gather into the same rows
        DgTransposeArgs t;
        memset(&t, 0, sizeof(t));
        t.nmaps = 4;
        t.src[0] = orig_feats; t.dst[0] = F32(p.nhwc_f[0]); t.K[0] = p.C; t.K4[0] = p.C4; t.HW[0] = p.h * p.w;
        t.src[1] = orig_feats_pos; t.dst[1] = F32(p.nhwc_f[1]); t.K[1] = p.C; t.K4[1] = p.C4; t.HW[1] = p.h * p.w;
        t.src[2] = orig_code; t.dst[2] = F32(p.nhwc_c[0]); t.K[2] = p.D; t.K4[2] = p.D4; t.HW[2] = p.hc * p.wc;
        t.src[3] = orig_code_pos; t.dst[3] = F32(p.nhwc_c[1]); t.K[3] = p.D; t.K4[3] = p.D4; t.HW[3] = p.hc * p.wc;
        DG_HIP(dg_launch_transpose(t, p.B, stream));
        DgGatherRowsArgs g;
        memset(&g, 0, sizeof(g));
        int nj = 0;
        for (int o = 0; o < p.nops; ++o) {
            const int srcsel = o == 1 ? 1 : 0;          // op 1 reads the *_pos maps, negatives read orig_feats / orig_code
            const float* coords = o == 0 ? coords1 : coords2;
            const int64_t* idx = o >= 2 ? perms + (size_t)(o - 2) * p.B : nullptr;
            g.src[nj] = F32(p.nhwc_f[srcsel]); g.coords[nj] = coords; g.srcidx[nj] = idx; g.rows[nj] = F32(p.rows_f[o]);
            g.K4[nj] = p.C4; g.Kout[nj] = (int)up(p.C, 128); g.as_bf16[nj] = 1; g.h[nj] = p.h; g.w[nj] = p.w; ++nj;
            g.src[nj] = F32(p.nhwc_c[srcsel]); g.coords[nj] = coords; g.srcidx[nj] = idx; g.rows[nj] = F32(p.rows_c[o]);
            g.K4[nj] = p.D4; g.Kout[nj] = p.D4; g.as_bf16[nj] = 0; g.h[nj] = p.hc; g.w[nj] = p.wc; ++nj;
        }
        g.njobs = nj; g.B = p.B; g.S = p.S; g.Sh = p.Sh; g.P = p.P;
        DG_HIP(dg_launch_gather_rows(g, stream));
    }
    DG_HIP(dg_launch_corr_small(a, stream));
    DG_HIP(dg_launch_small_finish(a, stream));
    return DG_OK;
}
struct FeatKeep { const float* keep[2]; float scale; };      // deferred Dropout2d of the two feature maps (dg_corr_forward_masked)

static int corr_forward_impl(const dg_corr_desc* desc, const float* orig_feats, const float* orig_feats_pos,
                             const float* orig_code, const float* orig_code_pos, const float* depth,
                             const float* coords1, const float* coords2, const int64_t* perms, const DrawArgs* draw,
                             float* out_scalars, void* workspace, size_t workspace_bytes, dg_stream_t stream_,
                             const FeatKeep* fk = nullptr, const float* feat_inv = nullptr);

extern "C" int dg_corr_forward(const dg_corr_desc* desc, const float* orig_feats, const float* orig_feats_pos,
                               const float* orig_code, const float* orig_code_pos, const float* depth,
                               const float* coords1, const float* coords2, const int64_t* perms,
                               float* out_scalars, void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    return corr_forward_impl(desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2, perms, nullptr,
                             out_scalars, workspace, workspace_bytes, stream_);
}

extern "C" int dg_corr_forward_extnorm(const dg_corr_desc* desc, const float* orig_feats, const float* orig_feats_pos,
                                       const float* orig_code, const float* orig_code_pos, const float* depth,
                                       const float* coords1, const float* coords2, const int64_t* perms, const float* feat_inv,
                                       float* out_scalars, void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (!feat_inv) return fail(DG_ERR_INVALID, "dg_corr_forward_extnorm: feat_inv is null");
    return corr_forward_impl(desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2, perms, nullptr,
                             out_scalars, workspace, workspace_bytes, stream_, nullptr, feat_inv);
}

extern "C" int dg_sampled_sumsq(int32_t B, int32_t C, int32_t h, int32_t w, int32_t S, int32_t line_grid, const float* feats,
                                const float* coords, const int64_t* srcidx, int32_t accumulate, float* out, dg_stream_t stream_) {
    if (B < 1 || C < 1 || h < 1 || w < 1 || S < 1 || !feats || !coords || !out) return fail(DG_ERR_INVALID, "dg_sampled_sumsq: bad arguments");
    DG_HIP(dg_launch_sampled_sumsq(feats, coords, srcidx, out, B, C, h, w, S, line_grid ? 1 : S, accumulate ? 1 : 0, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_corr_forward_draw(const dg_corr_desc* desc, const float* orig_feats, const float* orig_feats_pos,
                                    const float* orig_code, const float* orig_code_pos, const float* depth,
                                    const float* coords1, const float* coords2, int64_t* perms_out, uint64_t seed, void* perm_state,
                                    float* out_scalars, void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (!desc) return fail(DG_ERR_INVALID, "null descriptor");
    if (desc->n_neg > 0 && !perms_out) return fail(DG_ERR_INVALID, "perms_out is null with n_neg=%d", desc->n_neg);
    const DrawArgs draw{perms_out, seed, static_cast<unsigned long long*>(perm_state)};
    return corr_forward_impl(desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2, perms_out, &draw,
                             out_scalars, workspace, workspace_bytes, stream_);
}

extern "C" int dg_corr_forward_masked(const dg_corr_desc* desc, const float* orig_feats, const float* orig_feats_pos,
                                      const float* orig_code, const float* orig_code_pos, const float* depth,
                                      const float* coords1, const float* coords2, int64_t* perms, int32_t draw_perms, uint64_t seed,
                                      void* perm_state, const float* feat_keep, const float* feat_pos_keep, float keep_scale,
                                      float* out_scalars, void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (!desc) return fail(DG_ERR_INVALID, "null descriptor");
    if (desc->n_neg > 0 && !perms) return fail(DG_ERR_INVALID, "perms is null with n_neg=%d", desc->n_neg);
    if ((feat_keep || feat_pos_keep) && !(keep_scale > 0.f)) return fail(DG_ERR_INVALID, "keep_scale %g with keep flags", (double)keep_scale);
    const DrawArgs draw{perms, seed, static_cast<unsigned long long*>(perm_state)};
    const FeatKeep fk{{feat_keep, feat_pos_keep}, keep_scale};
    return corr_forward_impl(desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2, perms,
                             draw_perms ? &draw : nullptr, out_scalars, workspace, workspace_bytes, stream_, &fk);
}

static int corr_forward_impl(const dg_corr_desc* desc, const float* orig_feats, const float* orig_feats_pos,
                             const float* orig_code, const float* orig_code_pos, const float* depth,
                             const float* coords1, const float* coords2, const int64_t* perms, const DrawArgs* draw,
                             float* out_scalars, void* workspace, size_t workspace_bytes, dg_stream_t stream_, const FeatKeep* fk,
                             const float* feat_inv) {
    Plan p;
    int rc = make_plan(desc, p);
    if (rc != DG_OK) return rc;
    if (feat_inv && (p.ident || p.small))
        return fail(DG_ERR_INVALID, "dg_corr_forward_extnorm is the sampled-coordinate path above 160 positions: the identity grid takes "
                                    "DG_FEATS_UNIT, smaller grids any width as they are");
    if (fk && (fk->keep[0] || fk->keep[1]) && !p.ident)
        return fail(DG_ERR_UNSUPPORTED, "deferred feature dropout (feat_keep) is built for the identity grid only: "
                                        "with sampled coordinates hand the dropped features in");
    if (!orig_feats || !orig_feats_pos || !orig_code || !orig_code_pos || !coords1 || !coords2 || !out_scalars || !workspace)
        return fail(DG_ERR_INVALID, "null tensor pointer");
    if (p.N > 0 && !perms) return fail(DG_ERR_INVALID, "perms is null with n_neg=%d", p.N);
    if (p.depth && (!depth || desc->depth_h < 1 || desc->depth_w < 1))
        return fail(DG_ERR_INVALID, "DG_DEPTH_TERM set but depth is missing (the reference raises on depth=None too)");
    if (workspace_bytes < p.total) return fail(DG_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, p.total);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* ws = static_cast<char*>(workspace);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };

    // Exact clamp masks on the dense grid (round 6): the mask words depend on the code maps and the batch maps only, so their chain
    // (code norms + draw -> code operands -> k_cd_mask3) runs on the library's side stream BESIDE the feature side of the preparation
    // (k_prep_dense's feats, k_colmean, k_rowmean) instead of in front of the fused kernel: the same launches split by role, joined
    // in front of the fused kernel.  Without a side stream (first call inside a capture) everything runs in sequence as before.
    // (the region object - it holds the side stream's lock - exists only on calls that use it; the k_gs launches below re-use it)
    std::optional<SideRegion> side_region;
    if (p.xmask_dense && p.ident && split_masks_enabled()) side_region.emplace(stream);
    const bool split = side_region && static_cast<bool>(*side_region) && p.pointwise;
    SideRegion* const sidep = side_region ? &*side_region : nullptr;      // (only touched under `split`)
    DgPreArgs pre_late;
    memset(&pre_late, 0, sizeof(pre_late));
    // 1.+2. operands.  Identity grid: one launch builds both feats and both code operands straight from NCHW (+ the
    //       depth indicators).  General coords: channel-last copies (the gather reads whole channel vectors per tap), then
    //       sample + normalise + operand blobs.
    if (p.ident) {
        DgDenseArgs g;
        memset(&g, 0, sizeof(g));
        g.src[0] = orig_feats; g.src[1] = orig_feats_pos; g.code[0] = orig_code; g.code[1] = orig_code_pos;
        for (int o = 0; o < 2; ++o) { g.blob[o] = ws + p.op[o]; g.colpart[o] = F32(p.colpart[o]); g.inv_norm[o] = F32(p.inv[o]); g.ccolpart[o] = F32(p.ccolpart[o]); }
        g.depth = p.depth ? depth : nullptr; g.nz = F32(p.nz); g.nzsum = F32(p.nzsum);
        g.B = p.B; g.K = p.C; g.D = p.D; g.KF = p.KF; g.KD = p.KD; g.h = p.h; g.w = p.w; g.P = p.P; g.Ppad = p.Ppad;
        g.dH = desc->depth_h; g.dW = desc->depth_w;
        g.code_split = p.pointwise ? 1 : 0;        // (the code column sums then ride in the k_rowmean launch, which only pointwise has)
        g.unit = (desc->flags & DG_FEATS_UNIT) ? 1 : 0;
        if (fk) { g.fkeep[0] = fk->keep[0]; g.fkeep[1] = fk->keep[1]; g.fscale = fk->scale; }
        if (draw && p.N > 0) { g.draw_out = draw->out; g.draw_state = draw->state; g.draw_seed = draw->seed; g.draw_count = p.N; }
        if (split) {
            // the code roles (norms) and the draw on the side stream - the chain code norms -> code operands -> mask words hangs
            // off them and runs beside the feature operands, their means and the row means
            DgDenseArgs gs = g;
            gs.roles = 2 | 8; g.roles = 1 | 4;
            DG_HIP(sidep->fork());
            DG_HIP(dg_launch_prep_dense(gs, sidep->stream()));
        }
        DG_HIP(dg_launch_prep_dense(g, stream));
        if (p.xmask_dense && !p.pointwise) {
            // exact clamp masks without `pointwise` (the code operands are then built by k_prep_dense itself, without the parts the
            // split-fp16 form below needs): position-major fp32 rows of the two code maps (on this grid position p IS pixel p), then
            // the sign of every raw fp32 dot product (k_cd_mask) as one word per (S tile, R position) for all pair-sets
            DgTransposeArgs t;
            memset(&t, 0, sizeof(t));
            t.nmaps = 2;
            t.src[0] = orig_code; t.dst[0] = F32(p.nhwc_c[0]); t.K[0] = p.D; t.K4[0] = p.D4; t.HW[0] = p.h * p.w;
            t.src[1] = orig_code_pos; t.dst[1] = F32(p.nhwc_c[1]); t.K[1] = p.D; t.K4[1] = p.D4; t.HW[1] = p.h * p.w;
            DG_HIP(dg_launch_transpose(t, p.B, stream));
            DgCdMaskArgs m;
            memset(&m, 0, sizeof(m));
            m.rowsR = F32(p.nhwc_c[0]);
            for (int tt = 0; tt < p.T; ++tt) {
                m.rowsS[tt] = F32(p.nhwc_c[op_of(p, tt)]); m.sidx[tt] = map_of(p, tt, perms);
                m.bits[tt] = reinterpret_cast<uint32_t*>(ws + p.maskbits[tt]);
            }
            m.T = p.T; m.B = p.B; m.P = p.P; m.Ppad = p.Ppad; m.D = p.D; m.D4 = p.D4;
            DG_HIP(dg_launch_cd_mask(m, stream));
        }
    } else if (p.small) {
        return forward_small(p, desc, orig_feats, orig_feats_pos, orig_code, orig_code_pos, depth, coords1, coords2, perms, draw,
                             out_scalars, ws, stream);
    } else {
        // general coordinates: the first launch already reads the batch maps, so they are drawn by a launch of their own - which
        // also carries the other jobs that depend on nothing but the call's inputs: the depth indicators and, on gradient passes,
        // the inverse tap records of the sample() adjoint (dg_corr_backward then finds them in the workspace)
        {
            DgPreArgs q;
            memset(&q, 0, sizeof(q));
            if (draw && p.N > 0) { q.seed = draw->seed; q.state = draw->state; q.perms = draw->out; q.count = p.N; }
            if (p.depth) { q.depth = depth; q.nz = F32(p.nz); q.nzsum = F32(p.nzsum); q.dH = desc->depth_h; q.dW = desc->depth_w; }
            if (p.grad && (size_t)p.hc * p.wc <= 4096 && p.P <= 65535) { q.coords1 = coords1; q.coords2 = coords2; q.taps = ws + p.taps; }
            q.B = p.B; q.h = p.hc; q.w = p.wc; q.S = p.S; q.Sh = p.Sh; q.P = p.P; q.Ppad = p.Ppad;      // (h, w): the maps the tap records index = the code maps
            if (p.B > 8192 && q.count > 0) return fail(DG_ERR_UNSUPPORTED, "B=%d too large for the in-call draw", p.B);
            // Nothing in front of the fused kernel reads the depth indicators or the tap records: they ride as extra blocks of the
            // gather launch below (round 4); only the draw - the sampler's first input - keeps a launch of its own
            pre_late = q;
            pre_late.count = 0;
            q.depth = nullptr; q.taps = nullptr;
            DG_HIP(dg_launch_pre_general(q, stream));
        }
        if (p.rows) {
            DgPlaneArgs t;
            memset(&t, 0, sizeof(t));
            t.src[0] = orig_feats; t.src[1] = orig_feats_pos; t.src[2] = orig_code; t.src[3] = orig_code_pos;
            t.K[0] = t.K[1] = p.C; t.K4[0] = t.K4[1] = p.C4; t.K[2] = t.K[3] = p.D; t.K4[2] = t.K4[3] = p.D4;
            for (int o = 0; o < p.nops; ++o) { t.rows[o][0] = F32(p.rows_f[o]); t.rows[o][1] = F32(p.rows_c[o]); }
            t.coords1 = coords1; t.coords2 = coords2; t.perms = perms;
            t.nops = p.nops; t.B = p.B; t.h = p.h; t.w = p.w; t.S = p.S; t.Sh = p.Sh; t.P = p.P;
            DG_HIP(dg_launch_plane_sample(t, stream));
        } else {
            DgTransposeArgs t;
            memset(&t, 0, sizeof(t));
            t.nmaps = 4;
            t.src[0] = orig_feats; t.dst[0] = F32(p.nhwc_f[0]); t.K[0] = p.C; t.K4[0] = p.C4; t.HW[0] = p.h * p.w;
            t.src[1] = orig_feats_pos; t.dst[1] = F32(p.nhwc_f[1]); t.K[1] = p.C; t.K4[1] = p.C4; t.HW[1] = p.h * p.w;
            t.src[2] = orig_code; t.dst[2] = F32(p.nhwc_c[0]); t.K[2] = p.D; t.K4[2] = p.D4; t.HW[2] = p.hc * p.wc;
            t.src[3] = orig_code_pos; t.dst[3] = F32(p.nhwc_c[1]); t.K[3] = p.D; t.K4[3] = p.D4; t.HW[3] = p.hc * p.wc;
            DG_HIP(dg_launch_transpose(t, p.B, stream));
        }
        DgGatherArgs g;
        memset(&g, 0, sizeof(g));
        g.B = p.B; g.S = p.S; g.Sh = p.Sh; g.P = p.P; g.Ppad = p.Ppad; g.KF = p.KF; g.KD = p.KD;
        int nj = 0;
        for (int o = 0; o < p.nops; ++o) {
            const int srcsel = o == 1 ? 1 : 0;          // op 1 reads the *_pos maps, negatives read orig_feats/orig_code
            const float* coords = o == 0 ? coords1 : coords2;
            const int64_t* idx = o >= 2 ? perms + (size_t)(o - 2) * p.B : nullptr;
            DgGatherJob& f = g.jobs[nj++];
            f.src = p.rows ? F32(p.rows_f[o]) : F32(p.nhwc_f[srcsel]); f.coords = coords; f.srcidx = p.rows ? nullptr : idx;
            f.blob = ws + p.op[o]; f.inv_norm = nullptr; f.colpart = F32(p.colpart[o]);
            f.ext_inv = feat_inv ? feat_inv + (size_t)o * p.B * p.P : nullptr;
            f.K = p.C; f.K4 = p.C4; f.Kpad = p.KF; f.is_code = 0; f.h = p.h; f.w = p.w;
            DgGatherJob& c = g.jobs[nj++];
            c.src = p.rows ? F32(p.rows_c[o]) : F32(p.nhwc_c[srcsel]); c.coords = coords; c.srcidx = p.rows ? nullptr : idx;
            c.blob = ws + p.op[o]; c.inv_norm = F32(p.inv[o]); c.colpart = F32(p.ccolpart[o]);
            c.K = p.D; c.K4 = p.D4; c.Kpad = p.KD; c.is_code = 1; c.h = p.hc; c.w = p.wc;
        }
        g.njobs = nj;
        g.direct = p.rows ? 1 : 0;
        if (p.xmask) {
            // exact clamp masks of the small sample grids: the sign of every fp32 code dot product from the sampled rows
            // (k_plane_sample's, like the gather's input) - extra blocks of the gather launch (its own launch until round 4)
            DgCdMaskArgs& m = g.cd;
            m.rowsR = F32(p.rows_c[0]);
            for (int t = 0; t < p.T; ++t) {
                m.rowsS[t] = F32(p.rows_c[op_of(p, t)]); m.sidx[t] = map_of(p, t, perms);
                m.bits[t] = reinterpret_cast<uint32_t*>(ws + p.maskbits[t]);
            }
            m.T = p.T; m.B = p.B; m.P = p.P; m.Ppad = p.Ppad; m.D = p.D; m.D4 = p.D4;
        }
        g.pre = pre_late;
        DG_HIP(dg_launch_gather(g, p.KF, stream));
    }

    // (the launch plan of step 4 is needed here already: the consumer lists of k_corr2's grouped ragged blocks are written by
    //  extra blocks of the k_colmean launch)
    DgCorrArgs a;
    int depth_index;
    const int njA = build_corr_jobs(p, desc, ws, perms, a, &depth_index);

    // 3. column sums of the operands (mean feats for the centering, code sums for the cd means), then the row means of
    //    fd (pointwise centering as a rank-1 correction)
    {
        DgColmeanArgs c;
        memset(&c, 0, sizeof(c));
        c.nops = p.nops; c.B = p.B; c.P = p.P; c.Ppad = p.Ppad; c.KF = p.KF; c.KD = p.KD;
        for (int o = 0; o < p.nops; ++o) {
            c.colpart[o] = p.pointwise ? F32(p.colpart[o]) : nullptr; c.bbar[o] = F32(p.bbar[o]);
            c.bsplit[o] = reinterpret_cast<__bf16*>(ws + p.bsplit[o]);
            c.ngroups[o] = p.Ppad / 32;              // feats partial column sums: one group per tile on both paths
            c.ccolpart[o] = F32(p.ccolpart[o]); c.csum[o] = F32(p.csum[o]);
        }
        c.zero_word = (p.grad && p.depth) ? reinterpret_cast<unsigned int*>(ws + p.ticket) : nullptr;
        c.zero_words9 = a.wctr;
        if (a.gr_list) {           // the consumer lists of k_corr2's grouped ragged blocks ride along (extra blocks of this launch)
            const int nd = a.jobs[a.njobs - 1].kind == DG_JOB_DEPTH ? 1 : 0;
            c.gr.nh = a.njobs - nd; c.gr.nkeys = a.gr_nkeys; c.gr.B = p.B;
            for (int j = 0; j < c.gr.nh; ++j) { c.gr.sidx[j] = a.jobs[j].sidx; c.gr.key[j] = a.gr_key[j]; }
            c.gr.list = const_cast<int32_t*>(a.gr_list); c.gr.count = const_cast<int32_t*>(a.gr_count); c.gr.rank = const_cast<int16_t*>(a.gr_rank);
        }
        if (p.ident && p.pointwise) {         // dense code operands from channel planes (norms: k_prep_dense; csum: k_rowmean launch)
            c.dc.code[0] = orig_code; c.dc.code[1] = orig_code_pos;
            for (int o = 0; o < 2; ++o) { c.dc.blob[o] = ws + p.op[o]; c.dc.inv_norm[o] = F32(p.inv[o]); c.dc.ccolpart[o] = F32(p.ccolpart[o]); }
            c.dc.B = p.B; c.dc.D = p.D; c.dc.KF = p.KF; c.dc.KD = p.KD; c.dc.h = p.h; c.dc.w = p.w; c.dc.P = p.P; c.dc.Ppad = p.Ppad;
            if (p.xmask_dense) { c.dc.clo[0] = ws + p.clo[0]; c.dc.clo[1] = ws + p.clo[1]; }
        }
        if (split) {
            DgColmeanArgs cs = c;
            cs.zsel = 1; c.zsel = 2;
            DG_HIP(sidep->hand_over(0));                        // the draw (and the norms): the consumer lists below read the batch maps
            DG_HIP(dg_launch_colmean(cs, sidep->stream()));     // code operands (+ what the fp16 C parts drop)
        }
        DG_HIP(dg_launch_colmean(c, stream));
    }
    if (p.xmask_dense && p.pointwise) {
        // exact clamp masks: cd from split fp16 operands (the C parts + what they drop, both written by the launch above)
        DgCdMask3Args m;
        memset(&m, 0, sizeof(m));
        m.opR = ws + p.op[0]; m.loR = ws + p.clo[0];
        for (int tt = 0; tt < p.T; ++tt) {
            m.opS[tt] = ws + p.op[op_of(p, tt)]; m.loS[tt] = ws + p.clo[op_of(p, tt)]; m.sidx[tt] = map_of(p, tt, perms);
            m.bits[tt] = reinterpret_cast<uint32_t*>(ws + p.maskbits[tt]);
        }
        const DgBlob bl(p.KF, p.KD);
        m.T = p.T; m.B = p.B; m.Ppad = p.Ppad; m.blob_bytes = bl.bytes; m.off_c = bl.off_c; m.KD = p.KD;
        if (split) {
            DG_HIP(sidep->hand_over(1));                        // the code operands: k_rowmean reduces their column sums and writes FOLD's
                                                              // stash into the padding they zeroed
            DG_HIP(dg_launch_cd_mask3(m, sidep->stream()));
            DG_HIP(sidep->record_join());
        } else {
            DG_HIP(dg_launch_cd_mask3(m, stream));
        }
    }
    if (p.pointwise) {
        DgRowmeanArgs r;
        memset(&r, 0, sizeof(r));
        r.B = p.B; r.P = p.P; r.Ppad = p.Ppad; r.KF = p.KF; r.KD = p.KD; r.njobs = p.T; r.abar = F32(p.bbar[0]);
        for (int t = 0; t < p.T; ++t) {
            r.jobs[t].A = ws + p.op[0]; r.jobs[t].aidx = nullptr;
            r.jobs[t].bbar = F32(p.bbar[op_of(p, t)]); r.jobs[t].bidx = map_of(p, t, perms);
            r.jobs[t].bsplit = reinterpret_cast<const __bf16*>(ws + p.bsplit[op_of(p, t)]);
            r.jobs[t].rvec = F32(p.rvec[t]); r.jobs[t].rimg = F32(p.rimg[t]);
        }
        if (p.ident) {
            r.ncs = 2;
            for (int o = 0; o < 2; ++o) { r.cs_part[o] = F32(p.ccolpart[o]); r.cs_out[o] = F32(p.csum[o]); }
        }
        if (p.fold) { r.stash = ws + p.op[0]; r.stash_off = FOLD_STASH_OFF; }
        DG_HIP(dg_launch_rowmean(r, stream));
    }

    // 4. fused correlation passes
    if (split) DG_HIP(sidep->join());                           // the mask words
    DG_HIP(launch_main(p, a, njA, depth_index, stream));

    // 5. scalar outputs: the partial sums are reduced by the next launch (k_gs on a gradient pass)
    DgFinishArgs f;
    memset(&f, 0, sizeof(f));
    for (int j = 0; j < a.njobs; ++j) {
        f.part[j] = a.jobs[j].part; f.slot_loss[j] = a.jobs[j].slot_loss; f.slot_cd[j] = a.jobs[j].slot_cd;
        f.scale[j] = a.jobs[j].fin_scale;
    }
    f.njobs = a.njobs; f.nblk = p.B * p.nrb; f.B = p.B; f.P = p.P;
    const int dep_nrb = (p.Ppad / 32 + 7) / 8;               // row blocks of the depth term inside the k_gs launch: 8 row tiles each
    if (p.grad && p.depth) {                                 // its partial sums: one more entry of the reduction
        const int j = f.njobs++;
        f.part[j] = F32(p.part[p.T]); f.slot_loss[j] = DG_OUT_LOSS_DEPTH; f.slot_cd[j] = -1;
        f.scale[j] = (float)(1.0 / ((double)p.B * p.P * p.P));
        f.nblk_job[j] = p.B * dep_nrb;
    }
    f.nzsum = p.depth ? F32(p.nzsum) : nullptr;
    f.out = out_scalars;
    f.wtot[0] = desc->w_intra; f.wtot[1] = desc->w_inter; f.wtot[2] = desc->w_neg; f.wtot[3] = desc->w_depth;
    if (p.grad) {
        DgGsArgs g;
        const uint32_t* dep_maskbits = nullptr;
        build_gs_jobs(p, ws, perms, g);
        g.fin = f;
        if (p.depth) {
            g.dep_op = ws + p.op[0]; g.dep_nz = F32(p.nz); g.dep_dR = F32(p.dRA[p.T]); g.dep_part = F32(p.part[p.T]);
            g.dep_ticket = reinterpret_cast<unsigned int*>(ws + p.ticket);
            dep_maskbits = (p.xmask || p.xmask_dense) ? reinterpret_cast<const uint32_t*>(ws + p.maskbits[0]) : nullptr;   // cd of the depth term = intra's cd
            g.dep_shift = desc->shift_depth; g.dep_nrb = dep_nrb; g.dep_blocks = p.B * dep_nrb;
            clamp_bounds(desc, g.dep_lo, g.dep_hi);
        }
        if (p.xmask_dense && p.depth) {
            // the G-stream blocks as a launch of the plain kernel (two blocks per CU) and the depth blocks alone in the masked form
            // (256 registers per wave; the last depth block reduces the call's partial sums) - SIDE BY SIDE on a second stream
            // where the library has one (fork / join by events, capturable into a hipGraph): the stream launch is bound by HBM
            // bytes, the 128 depth blocks by a latency chain on half the CUs (one behind the other: 76 + 52 us)
            DgGsArgs gstream = g;
            gstream.dep_blocks = 0; gstream.fin.out = nullptr;
            if (!side_region) side_region.emplace(stream);
            SideRegion& side = *side_region;
            side.reset();                                     // (a second fork .. join of the same region object)
            if (side) {
                DG_HIP(side.fork());
                DG_HIP(dg_launch_gs(g, dep_maskbits, side.stream(), true));
                DG_HIP(side.record_join());
                DG_HIP(dg_launch_gs(gstream, nullptr, stream, false, p.half));
                DG_HIP(side.join());
            } else {
                DG_HIP(dg_launch_gs(gstream, nullptr, stream, false, p.half));
                DG_HIP(dg_launch_gs(g, dep_maskbits, stream, true));
            }
        } else {
            DG_HIP(dg_launch_gs(g, dep_maskbits, stream, false, p.half));
        }
    } else {
        DG_HIP(dg_launch_finish(f, stream));
    }

    return DG_OK;
}

static int corr_backward_impl(const dg_corr_desc* desc, const float* grad_scalars, const float* grad_total, const float* coords1,
                              const float* coords2, const int64_t* perms, float* grad_code, float* grad_code_pos,
                              void* workspace, size_t workspace_bytes, dg_stream_t stream_);

extern "C" int dg_corr_backward(const dg_corr_desc* desc, const float* grad_scalars, const float* coords1,
                                const float* coords2, const int64_t* perms, float* grad_code, float* grad_code_pos,
                                void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (!grad_scalars) return fail(DG_ERR_INVALID, "null pointer");
    return corr_backward_impl(desc, grad_scalars, nullptr, coords1, coords2, perms, grad_code, grad_code_pos, workspace,
                              workspace_bytes, stream_);
}

extern "C" int dg_corr_backward_total(const dg_corr_desc* desc, const float* grad_total, const float* coords1,
                                      const float* coords2, const int64_t* perms, float* grad_code, float* grad_code_pos,
                                      void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (!grad_total) return fail(DG_ERR_INVALID, "null pointer");
    return corr_backward_impl(desc, nullptr, grad_total, coords1, coords2, perms, grad_code, grad_code_pos, workspace,
                              workspace_bytes, stream_);
}

static int corr_backward_impl(const dg_corr_desc* desc, const float* grad_scalars, const float* grad_total, const float* coords1,
                              const float* coords2, const int64_t* perms, float* grad_code, float* grad_code_pos,
                              void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    Plan p;
    int rc = make_plan(desc, p);
    if (rc != DG_OK) return rc;
    if (!p.grad) return fail(DG_ERR_INVALID, "dg_corr_backward needs a descriptor with DG_NEED_GRAD (as used in forward)");
    if (!coords1 || !coords2 || !grad_code || !grad_code_pos || !workspace) return fail(DG_ERR_INVALID, "null pointer");
    if (p.N > 0 && !perms) return fail(DG_ERR_INVALID, "perms is null with n_neg=%d", p.N);
    if (workspace_bytes < p.total) return fail(DG_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, p.total);
    char* ws = static_cast<char*>(workspace);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    DgScatterArgs s;
    memset(&s, 0, sizeof(s));
    // the kernels keep -G (mask * (fd'' - shift)) and its products: the sign lives here
    const float f = (float)(-1.0 / ((double)p.B * p.P * p.P));
    const float fn = p.N > 0 ? f / (float)p.N : 0.f;
    int n = 0;
    auto add = [&](size_t buf, const int64_t* route, int gidx, int csel, float factor, int dest, int raw, int half = 0) {
        s.src[n].buf = F32(buf); s.src[n].route = route; s.src[n].gidx = gidx; s.src[n].coords_sel = csel;
        s.src[n].factor = factor; s.src[n].dest = dest; s.src[n].raw = raw; s.src[n].half = half; ++n;
    };
    const int hf = p.half ? 1 : 0;              // (the pair-sets' tiles of k_corr2 and k_gs; the depth term's stay fp32)
    // dRA[t]: the fused kernel's raw accumulator-order tiles (stationary operand = operand 1 for every pair-set);
    // dRB[t]: k_gs output, row-major, normalisation backward already applied
    if (p.small) {
        // the fused small-grid kernel (dg_small.hip): stationary-side tiles raw, streamed-side tiles final (one set per half of the
        // stationary tiles), and with `pointwise` the same again for the old_mean term, whose factor old_mean_t lives on the device
        for (int t = 0; t < p.T; ++t) {
            const int gi = t < 2 ? t : 2;
            const float ft = t < 2 ? f : fn;
            const int64_t* route = t >= 2 ? perms + (size_t)(t - 2) * p.B : nullptr;
            const int csel = t == 0 ? 0 : 1, dest = t == 1 ? 1 : 0;
            add(p.dRA[t], nullptr, gi, 0, ft, 0, 1);
            const float* om = p.pointwise ? F32(p.om) + t : nullptr;
            if (p.pointwise) { add(p.dRA2[t], nullptr, gi, 0, ft, 0, 1); s.src[n - 1].dfac = om; }
            // the streamed-side (final) tiles: one set per half of the stationary tiles, with `pointwise` the old_mean terms on top.
            // ROUTED sources (the negatives) are merged into one buffer each by extra slices of the combine launch, in front of the
            // adjoint launch that reads the result - routed sources are what that launch's time scales with.  Direct ones (intra,
            // inter) are read by the combine launch itself: those keep their terms.
            if (route && (p.pointwise || p.nsplit == 2)) {
                const int j = s.naxpy++;
                s.axo[j] = F32(p.dRBm[t]); s.axd[j] = F32(p.dRB[t]); s.axd2[j] = p.nsplit == 2 ? F32(p.dRBs[t]) : nullptr;
                s.axs[j] = p.pointwise ? F32(p.dRB2[t][0]) : nullptr; s.axs2[j] = (p.pointwise && p.nsplit == 2) ? F32(p.dRB2[t][1]) : nullptr;
                s.axf[j] = om;
                add(p.dRBm[t], route, gi, csel, ft, dest, 0);
            } else {
                for (int k = 0; k < p.nsplit; ++k) {
                    add(k == 0 ? p.dRB[t] : p.dRBs[t], route, gi, csel, ft, dest, 0);
                    if (p.pointwise) { add(p.dRB2[t][k], route, gi, csel, ft, dest, 0); s.src[n - 1].dfac = om; }
                }
            }
        }
    } else {
    if (intra_is_symmetric(p)) add(p.dRA[0], nullptr, 0, 0, 2.0f * f, 0, 1, hf);      // -G symmetric: d/dc1 + d/dc2 = 2 d/dc1 (no k_gs job, build_gs_jobs)
    else { add(p.dRA[0], nullptr, 0, 0, f, 0, 1, hf); add(p.dRB[0], nullptr, 0, 0, f, 0, 0, hf); }
    add(p.dRA[1], nullptr, 1, 0, f, 0, 1, hf);
    add(p.dRB[1], nullptr, 1, 1, f, 1, 0, hf);
    for (int k = 0; k < p.N; ++k) {
        add(p.dRA[2 + k], nullptr, 2, 0, fn, 0, 1, hf);
        add(p.dRB[2 + k], perms + (size_t)k * p.B, 2, 1, fn, 0, 0, hf);
    }
    }
    if (p.depth) add(p.dRA[p.T], nullptr, 3, 0, 2.0f * f, 0, 1);   // dd and cd symmetric: d/dc1 + d/dc2 = 2 d/dc1
    s.nsrc = n;
    s.coords1 = coords1; s.coords2 = coords2; s.gscal = grad_scalars; s.gtot = grad_total;
    s.wtot[0] = desc->w_intra; s.wtot[1] = desc->w_inter; s.wtot[2] = desc->w_neg; s.wtot[3] = desc->w_depth;
    s.comb[0] = F32(p.comb[0]); s.comb[1] = F32(p.comb[1]);
    s.taps = ws + p.taps;
    {
        const DgBlob bl(p.KF, p.KD);
        s.xop = ws + p.op[0]; s.xinv = F32(p.inv[0]); s.blob_bytes = bl.bytes; s.blob_off_c = bl.off_c;
        // (half, final sources: destination 0 = the code map behind operand 0 - the negatives' streamed operand on the shared grid -,
        //  destination 1 = operand 1's)
        s.xinv_dest[0] = F32(p.inv[0]); s.xinv_dest[1] = p.nops > 1 ? F32(p.inv[1]) : nullptr;
    }
    s.out[0] = grad_code; s.out[1] = grad_code_pos;
    s.B = p.B; s.D = p.D; s.DP = p.KD; s.h = p.hc; s.w = p.wc; s.S = p.S; s.Sh = p.Sh; s.P = p.P; s.Ppad = p.Ppad;     // (h, w): the code maps
    if ((size_t)p.hc * p.wc > 4096) return fail(DG_ERR_UNSUPPORTED, "code map %dx%d too large for the gradient gather (max 4096 pixels)", p.hc, p.wc);
    s.DC = 8;
    s.dense = p.ident ? 1 : 0;
    s.taps_ready = p.ident ? 0 : 1;            // (general coordinates: built by the forward's first launch, dg_launch_pre_general)
    DG_HIP(dg_launch_scatter(s, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

static int materialize_impl(const dg_corr_desc* desc, int32_t which, const int64_t* perms, float* out_cd, float* out_loss,
                            void* workspace, size_t workspace_bytes, dg_stream_t stream_);

extern "C" int dg_corr_materialize(const dg_corr_desc* desc, int32_t which, float* out_cd, float* out_loss,
                                   void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    return materialize_impl(desc, which, nullptr, out_cd, out_loss, workspace, workspace_bytes, stream_);
}

extern "C" int dg_corr_materialize_shared(const dg_corr_desc* desc, int32_t which, const int64_t* perms, float* out_cd, float* out_loss,
                                          void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    return materialize_impl(desc, which, perms, out_cd, out_loss, workspace, workspace_bytes, stream_);
}

static int materialize_impl(const dg_corr_desc* desc, int32_t which, const int64_t* perms, float* out_cd, float* out_loss,
                            void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    Plan p;
    int rc = make_plan(desc, p);
    if (rc != DG_OK) return rc;
    if (!workspace) return fail(DG_ERR_INVALID, "null workspace");
    if (workspace_bytes < p.total) return fail(DG_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, p.total);
    if (which < -1 || which >= p.T) return fail(DG_ERR_INVALID, "which=%d outside [-1,%d)", which, p.T);
    if (which == -1 && !p.depth) return fail(DG_ERR_INVALID, "depth term not enabled in descriptor");
    if (which >= 2 && p.shared && !perms)
        return fail(DG_ERR_INVALID, "materialising a negative of a DG_SHARED_COORDS call needs its batch maps: dg_corr_materialize_shared");
    if (!out_cd && !out_loss) return DG_OK;
    char* ws = static_cast<char*>(workspace);
    if (p.small) {          // the fused small-grid kernel again, on the rows (and old_mean_t) the forward left in the workspace
        DgSmallArgs m;
        small_args(p, desc, ws, perms, m);
        m.mat = 1; m.mat_t = which; m.out_cd = out_cd; m.out_loss = out_loss; m.grad = 0; m.span = nullptr;
        DG_HIP(dg_launch_corr_small(m, static_cast<hipStream_t>(stream_)));
        return DG_OK;
    }
    // (a gradient pass with k_corr2's FOLD left the intra row means in the padding of the operand-1 blobs' C part: k_corr_main's
    //  un-reduced forms multiply all of it.  Cleared for this launch and written back behind it, from the row means that are still in
    //  the workspace: the workspace stays what the forward prepared - dg_corr_relaunch_main remains valid.)
    const int stash_off = (int)DgBlob(p.KF, p.KD).off_c + FOLD_STASH_OFF;
    if (p.fold)
        DG_HIP(dg_launch_set_stash(ws + p.op[0], p.B, p.Ppad / 32, (size_t)p.blob, stash_off, nullptr, p.P, p.Ppad, static_cast<hipStream_t>(stream_)));
    DgCorrArgs a;
    corr_args_base(p, desc, ws, a);
    // stationary = operand 2 (on MFMA lanes) -> the stores of one accumulator register are contiguous along q
    DgJob j = which == -1 ? depth_job(p, desc, ws) : helper_job(p, desc, ws, which, true, perms);
    j.center_on_lane = 0;
    j.out_cd = out_cd; j.out_loss = out_loss; j.part = nullptr; j.dR = nullptr;
    a.jobs[0] = j; a.njobs = 1;
    a.pos_w = p.ident ? p.w : 0;
    DG_HIP(dg_launch_corr(a, p.KF, p.KD, p.rf, 2, static_cast<hipStream_t>(stream_)));
    if (p.fold)
        DG_HIP(dg_launch_set_stash(ws + p.op[0], p.B, p.Ppad / 32, (size_t)p.blob, stash_off, reinterpret_cast<const float*>(ws + p.rvec[0]),
                                   p.P, p.Ppad, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

// Measurement aid: re-launch ONLY the fused correlation kernel on the operands a previous dg_corr_forward
// (same desc / perms / workspace) left in the workspace.  Idempotent (rewrites the same outputs).
extern "C" int dg_corr_relaunch_main(const dg_corr_desc* desc, const int64_t* perms, void* workspace,
                                     size_t workspace_bytes, dg_stream_t stream_) {
    Plan p;
    int rc = make_plan(desc, p);
    if (rc != DG_OK) return rc;
    if (!workspace || workspace_bytes < p.total) return fail(DG_ERR_WORKSPACE, "workspace missing or too small");
    if (p.N > 0 && !perms) return fail(DG_ERR_INVALID, "perms is null");
    if (p.small) {
        // (the scalars of the re-launch go to the workspace's scratch vector: the call's own outputs are not touched)
        DgSmallArgs m;
        small_args(p, desc, static_cast<char*>(workspace), perms, m);
        m.out = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.scratch_out);
        DG_HIP(dg_launch_corr_small(m, static_cast<hipStream_t>(stream_)));        // (the kernel alone: what the roofline leg times)
        return DG_OK;
    }
    DgCorrArgs a;
    int depth_index;
    const int njA = build_corr_jobs(p, desc, static_cast<char*>(workspace), perms, a, &depth_index);
    DG_HIP(launch_main(p, a, njA, depth_index, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_corr_intra_folded(const dg_corr_desc* desc) {
    Plan p;
    if (make_plan(desc, p) != DG_OK) return -1;
    return p.fold ? 1 : 0;
}

extern "C" const char* dg_corr_main_kernel_name(const dg_corr_desc* desc) {
    Plan p;
    if (make_plan(desc, p) != DG_OK) return nullptr;
    if (p.small) return "k_corr_small";
    // the job table holds addresses only: a made-up workspace base and batch-map pointer decide nothing but null / non-null
    char* const ws = reinterpret_cast<char*>(static_cast<uintptr_t>(1) << 21);
    const int64_t* const perms = reinterpret_cast<const int64_t*>(static_cast<uintptr_t>(1) << 20);
    DgCorrArgs a;
    int depth_index;
    const int njA = build_corr_jobs(p, desc, ws, perms, a, &depth_index);
    return (p.grad && njA > 0 && dg_corr2_supported(a, p.KF, p.KD)) ? "k_corr2" : "k_corr_main";
}

extern "C" int dg_super_perms(const float* keys, int32_t count, int32_t B, int64_t* out, dg_stream_t stream_) {
    if (count < 0 || B < 1 || B > 8192) return fail(DG_ERR_INVALID, "dg_super_perms: count=%d B=%d outside the supported range", count, B);
    if (count == 0) return DG_OK;
    if (!keys || !out) return fail(DG_ERR_INVALID, "null pointer");
    DG_HIP(dg_launch_super_perms(keys, 0ull, nullptr, count, B, out, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_super_perms_seeded(uint64_t seed, int32_t count, int32_t B, int64_t* out, dg_stream_t stream_) {
    if (count < 0 || B < 1 || B > 8192) return fail(DG_ERR_INVALID, "bad super_perm dimensions");
    if (count == 0) return DG_OK;
    if (!out) return fail(DG_ERR_INVALID, "null pointer");
    DG_HIP(dg_launch_super_perms(nullptr, seed, nullptr, count, B, out, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_super_perms_state(uint64_t* state, int32_t count, int32_t B, int64_t* out, dg_stream_t stream_) {
    if (count < 0 || B < 1 || B > 8192) return fail(DG_ERR_INVALID, "bad super_perm dimensions");
    if (count == 0) return DG_OK;
    if (!out || !state) return fail(DG_ERR_INVALID, "null pointer");
    DG_HIP(dg_launch_super_perms(nullptr, 0ull, reinterpret_cast<unsigned long long*>(state), count, B, out, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_rand_coords_state(uint64_t* state, int64_t n, float* out, dg_stream_t stream_) {
    if (!state || !out) return fail(DG_ERR_INVALID, "null pointer");
    if (n < 1 || n > (1ll << 24)) return fail(DG_ERR_INVALID, "dg_rand_coords_state: n=%lld outside [1, 2^24]", (long long)n);
    DG_HIP(dg_launch_rand_coords_state(reinterpret_cast<unsigned long long*>(state), out, (int)n, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_rand_keep_state(uint64_t* state, int64_t n, float p_keep, float* out, dg_stream_t stream_) {
    if (!state || !out) return fail(DG_ERR_INVALID, "null pointer");
    if (n < 1 || n > (1ll << 24) || !(p_keep >= 0.f && p_keep <= 1.f)) return fail(DG_ERR_INVALID, "dg_rand_keep_state: n=%lld, p_keep=%g", (long long)n, (double)p_keep);
    DG_HIP(dg_launch_rand_coords_state(reinterpret_cast<unsigned long long*>(state), out, (int)n, static_cast<hipStream_t>(stream_), p_keep));
    return DG_OK;
}

extern "C" int dg_salience_coords(const float* salience, int32_t B, int32_t H, int32_t W, int32_t n, const float* u_sel,
                                  const float* u_fallback, float* out_coords, dg_stream_t stream_) {
    if (!salience || !u_sel || !u_fallback || !out_coords) return fail(DG_ERR_INVALID, "null pointer");
    if (B < 1 || H < 1 || W < 1 || n < 1) return fail(DG_ERR_INVALID, "bad salience sampler dimensions");
    if ((size_t)H * W > (1u << 24)) return fail(DG_ERR_UNSUPPORTED, "salience map %dx%d too large", H, W);
    DG_HIP(dg_launch_salience_coords(salience, B, H, W, n, u_sel, u_fallback, out_coords, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_simple_depth_coords(const float* depth, int32_t B, int32_t depth_h, int32_t depth_w, int32_t h, int32_t w,
                                      int32_t n, const float* u_value, const float* u_pick, float* out_coords,
                                      dg_stream_t stream_) {
    if (!depth || !u_value || !u_pick || !out_coords) return fail(DG_ERR_INVALID, "null pointer");
    if (B < 1 || h < 1 || w < 1 || n < 1 || depth_h < 1 || depth_w < 1) return fail(DG_ERR_INVALID, "bad sampler dimensions");
    if ((size_t)h * w > 4096) return fail(DG_ERR_UNSUPPORTED, "feature map %dx%d too large for the sampler (max 4096 pixels)", h, w);
    DG_HIP(dg_launch_simple_coords(depth, B, depth_h, depth_w, h, w, n, u_value, u_pick, out_coords,
                                   static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_confusion_update(const int64_t* preds, const int64_t* target, int64_t count, int32_t n_classes,
                                   int32_t extra_clusters, int64_t* stats, dg_stream_t stream_) {
    if (count < 0 || n_classes < 1 || extra_clusters < 0) return fail(DG_ERR_INVALID, "bad confusion-matrix dimensions");
    if (count == 0) return DG_OK;
    if (!preds || !target || !stats) return fail(DG_ERR_INVALID, "null pointer");
    if ((long long)n_classes * (n_classes + extra_clusters) > (1 << 24)) return fail(DG_ERR_UNSUPPORTED, "confusion matrix too large");
    DG_HIP(dg_launch_confusion(reinterpret_cast<const long long*>(preds), reinterpret_cast<const long long*>(target), count,
                               n_classes, n_classes + extra_clusters, reinterpret_cast<unsigned long long*>(stats),
                               static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_knn_similarities(const float* queries, const float* feats, int64_t rows_q, int64_t n, int32_t F, int64_t q_stride,
                                   int64_t f_stride, float* out, int64_t out_stride, dg_stream_t stream_) {
    if (rows_q < 0 || n < 0 || F < 1 || q_stride < F || f_stride < F || out_stride < n) return fail(DG_ERR_INVALID, "bad similarity dimensions");
    if (rows_q == 0 || n == 0) return DG_OK;
    if (!queries || !feats || !out) return fail(DG_ERR_INVALID, "null pointer");
    if (n > 65535ll * 128) return fail(DG_ERR_UNSUPPORTED, "more than 8,388,480 candidate rows per call");
    DG_HIP(dg_launch_sims_nt(queries, feats, rows_q, n, F, q_stride, f_stride, out, out_stride, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_topk_rows(const float* vals, int64_t rows, int64_t cols, int64_t row_stride, int32_t k, int64_t* out_idx,
                            float* out_val, dg_stream_t stream_) {
    if (rows < 0 || cols < 1 || k < 1 || row_stride < cols) return fail(DG_ERR_INVALID, "bad top-k dimensions");
    if (k > 64 || k > cols) return fail(DG_ERR_UNSUPPORTED, "top-k needs k <= 64 and k <= cols (k=%d, cols=%lld)", k, (long long)cols);
    if (cols >= (1ll << 32) || rows >= (1ll << 31)) return fail(DG_ERR_UNSUPPORTED, "similarity matrix too large");
    if (rows == 0) return DG_OK;
    if (!vals || !out_idx) return fail(DG_ERR_INVALID, "null pointer");
    DG_HIP(dg_launch_topk_rows(vals, rows, cols, row_stride, k, reinterpret_cast<long long*>(out_idx), out_val,
                               static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

static int lhp_check(int32_t B, int32_t D, int32_t h, int32_t w) {
    if (B < 1 || D < 1 || h < 1 || w < 1) return fail(DG_ERR_INVALID, "bad LHP dimensions");
    if (D > 128) return fail(DG_ERR_UNSUPPORTED, "D=%d > 128 code channels not supported", D);
    if ((size_t)h * w > 4096) return fail(DG_ERR_UNSUPPORTED, "feature map %dx%d too large for the LHP propagation (max 4096 positions)", h, w);
    return DG_OK;
}

extern "C" int dg_lhp_forward(const float* code, const float* depth, int32_t B, int32_t D, int32_t h, int32_t w, int32_t depth_h,
                              int32_t depth_w, float* out, float* points, float* stats, dg_stream_t stream_) {
    if (int rc = lhp_check(B, D, h, w)) return rc;
    if (depth_h < 1 || depth_w < 1) return fail(DG_ERR_INVALID, "bad depth size");
    if (!code || !depth || !out || !points || !stats) return fail(DG_ERR_INVALID, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const uint32_t bits = 0x404f54cbu;            // 2*tan(90/2 rad), the reference's float32 factor (dg_fps_coords)
    float factor;
    memcpy(&factor, &bits, 4);
    DG_HIP(dg_launch_lhp_points(depth, B, depth_h, depth_w, h, w, factor, points, s));
    DG_HIP(dg_launch_lhp_propagate(false, code, points, stats, B, D, h * w, out, s));
    return DG_OK;
}

extern "C" int dg_lhp_backward(const float* grad_out, const float* points, const float* stats, int32_t B, int32_t D, int32_t h,
                               int32_t w, float* grad_code, dg_stream_t stream_) {
    if (int rc = lhp_check(B, D, h, w)) return rc;
    if (!grad_out || !points || !stats || !grad_code) return fail(DG_ERR_INVALID, "null pointer");
    DG_HIP(dg_launch_lhp_propagate(true, grad_out, points, const_cast<float*>(stats), B, D, h * w, grad_code,
                                   static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_lhp_map_forward(int32_t mode, const float* code, const float* attn, const float* depth, const float* divide,
                                  int32_t B, int32_t D, int32_t h, int32_t w, int32_t heads, int32_t depth_h, int32_t depth_w,
                                  float* out, float* map, float* points, dg_stream_t stream_) {
    if (int rc = lhp_check(B, D, h, w)) return rc;
    if (mode < DG_LHP_ATTN || mode > DG_LHP_ORIG_ATTN) return fail(DG_ERR_INVALID, "unknown LHP map mode %d", mode);
    if (!code || !out) return fail(DG_ERR_INVALID, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (mode == DG_LHP_ORIG_DEPTH) {
        if (!depth || !points || depth_h < 1 || depth_w < 1) return fail(DG_ERR_INVALID, "the depth map and the points scratch are required");
        const uint32_t bits = 0x404f54cbu;        // 2*tan(90/2 rad), as in dg_lhp_forward
        float factor;
        memcpy(&factor, &bits, 4);
        DG_HIP(dg_launch_lhp_points(depth, B, depth_h, depth_w, h, w, factor, points, s));
    } else if (!attn || heads < 1) {
        return fail(DG_ERR_INVALID, "the attention tensor (B,heads,h*w+1,h*w+1) is required");
    }
    if (!map) return fail(DG_ERR_INVALID, "the map buffer is required");
    if (mode != DG_LHP_ATTN && !divide) return fail(DG_ERR_INVALID, "divide_num is required");
    DG_HIP(dg_launch_lhp_map(mode, code, attn, points, divide, B, D, h, w, heads, out, map, s));
    return DG_OK;
}

extern "C" int dg_lhp_map_backward(int32_t mode, const float* grad_out, const float* map, const float* divide, int32_t B, int32_t D,
                                   int32_t h, int32_t w, float* grad_code, dg_stream_t stream_) {
    if (int rc = lhp_check(B, D, h, w)) return rc;
    if (mode < DG_LHP_ATTN || mode > DG_LHP_ORIG_ATTN) return fail(DG_ERR_INVALID, "unknown LHP map mode %d", mode);
    if (!grad_out || !map || !grad_code || (mode != DG_LHP_ATTN && !divide)) return fail(DG_ERR_INVALID, "null pointer");
    DG_HIP(dg_launch_lhp_map_bwd(mode, grad_out, map, divide, B, D, h, w, grad_code, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" size_t dg_fps_workspace_bytes(int32_t B, int32_t h, int32_t w) {
    // the pooled depth maps of B images (adaptive_avg_pool2d to the feature map, written by a launch over the whole chip in front of
    // the sampler); the sampler itself keeps its state in LDS
    if (B < 1 || h < 1 || w < 1) return 256;
    return (size_t)B * h * w * 4 + 256;
}

static int fps_entry(const float* depth, const float* depth_b, int32_t Ba, int32_t B, int32_t depth_h, int32_t depth_w, int32_t h,
                     int32_t w, int32_t S, float* out_coords, int32_t* out_inds, void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (!depth || !out_coords || (Ba < B && !depth_b)) return fail(DG_ERR_INVALID, "null pointer");
    if (B < 1 || Ba < 1 || h < 1 || w < 1 || S < 1 || depth_h < h || depth_w < w) return fail(DG_ERR_INVALID, "bad FPS dimensions");
    if (S * S > h * w) return fail(DG_ERR_INVALID, "cannot sample %d points from a %dx%d map", S * S, h, w);
    if ((size_t)h * w > 4096) return fail(DG_ERR_UNSUPPORTED, "feature map %dx%d too large for the sampler (max 4096 pixels)", h, w);
    // 2*tan(fov/2) with fov = 90 taken in radians (reference quirk, src/modules.py:989,1016), float32 bits
    const uint32_t bits = 0x404f54cbu;
    float factor;
    memcpy(&factor, &bits, 4);
    // (a workspace that is missing or too small is not an error: the sampler then pools inside its own blocks, one image per CU)
    float* pooled = (workspace && workspace_bytes >= (size_t)B * h * w * 4) ? static_cast<float*>(workspace) : nullptr;
    DG_HIP(dg_launch_fps(depth, depth_b, Ba, B, depth_h, depth_w, h, w, S, factor, out_coords, out_inds, pooled, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_fps_coords(const float* depth, int32_t B, int32_t depth_h, int32_t depth_w, int32_t h, int32_t w,
                             int32_t S, float* out_coords, int32_t* out_inds, void* workspace, size_t workspace_bytes,
                             dg_stream_t stream_) {
    return fps_entry(depth, nullptr, B, B, depth_h, depth_w, h, w, S, out_coords, out_inds, workspace, workspace_bytes, stream_);
}

extern "C" int dg_fps_coords_pair(const float* depth, const float* depth_pos, int32_t B, int32_t depth_h, int32_t depth_w,
                                  int32_t h, int32_t w, int32_t S, float* out_coords, int32_t* out_inds, void* workspace,
                                  size_t workspace_bytes, dg_stream_t stream_) {
    return fps_entry(depth, depth_pos, B, 2 * B, depth_h, depth_w, h, w, S, out_coords, out_inds, workspace, workspace_bytes, stream_);
}

// ---- the segmentation head and the probes (dg_head.hip, dg_probe.hip)
static int head_check(int32_t B, int32_t C, int32_t D, int32_t P) {
    if (B < 1 || C < 1 || D < 1 || P < 1) return fail(DG_ERR_INVALID, "bad head dimensions");
    if (C > 768 || (C & 7)) return fail(DG_ERR_UNSUPPORTED, "C=%d: the head needs C <= 768 and a multiple of 8", C);
    if (D > 128) return fail(DG_ERR_UNSUPPORTED, "D=%d > 128 code channels not supported", D);
    return DG_OK;
}
static int head_splits(int32_t B, int32_t M, int32_t N, int32_t P, int32_t M2 = 0) {     // (M2: a second product in the same launch)
    const int steps = B * ((P + 31) / 32);
    if (dg_head_wgrad_one_pass(M, N, M2, P) && steps >= 8) {
        // k_head_wgrad3: one block per CU and one round - the largest multiple of 8 splits whose blocks (channel tiles x splits) fit 256 CUs
        const int s = (256 / ((N + 127) / 128)) & ~7;
        return s > steps ? (steps & ~7) : s;              // (a multiple of 8 either way: what the launcher checks)
    }
    const int tiles = ((M + 127) / 128 + (M2 + 127) / 128) * ((N + 127) / 128);
#ifndef HEAD_SPLIT_TARGET
    // about two blocks per CU; three when both featurizer passes of a step share the launch (1600 position steps at the headline:
    // 512 / 640 / 768 / 896 blocks gave 136 / 132 / 128 / 133 us for the two weight-gradient launches and their reduction)
    const int HEAD_SPLIT_TARGET = steps >= 1200 ? 768 : 512;
#endif
    int s = (HEAD_SPLIT_TARGET + tiles - 1) / tiles;
    s = (s + 7) & ~7;                                       // a multiple of 8: k_head_wgrad2 keeps the tiles of a split on one XCD
    return s > steps ? steps : s;
}
struct HeadPlan { size_t dh, p2a, p1, p2b, pbd, pb2a, gbf, total; int s2a, s1, s2b, tiles, dh_blocks; bool one_pass; };
static HeadPlan head_plan(int32_t B, int32_t C, int32_t D, int32_t P) {
    HeadPlan h;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += up(bytes, 256); return o; };
    // cluster2: d W2a and d W1 share the feature operand and one launch (same splits); cluster1 alone (linear head): d W1 by itself
    h.s2a = head_splits(B, C, C, P, D); h.s1 = head_splits(B, D, C, P); h.s2b = h.s1;
#ifdef DG_DEVTOOLS
    if (const char* e = getenv("DG_HEAD_S2B")) { const int v = atoi(e) & ~7; if (v >= 8 && v <= h.s2b) h.s2b = v; }
#endif
    // k_head_dh2 (the headline widths) forms d W2b beside d hidden: one partial sum per block of its launch
    h.dh_blocks = dg_head_dh_fused_blocks(B, C, D, P);
    if (h.dh_blocks > 0) h.s2b = h.dh_blocks;
    h.tiles = (P + 63) / 64;
    h.dh = take((size_t)B * C * ((P + 31) / 32 * 32) * 2);      // (+ the padding of the last step: the step-major form of k_head_dh2 / k_head_wgrad3)
    h.p2a = take((size_t)h.s2a * C * C * 4);
    h.p1 = take((size_t)(h.s1 > h.s2a ? h.s1 : h.s2a) * D * C * 4);
    h.p2b = take((size_t)h.s2b * D * C * 4);
    h.pbd = take((size_t)B * h.tiles * D * 4);
    h.pb2a = take((size_t)B * h.tiles * C * 4);
    // k_head_wgrad3 (d W2a and d W1 in one pass over the features) reads d code as the bf16 copy k_head_dh leaves
    h.one_pass = dg_head_wgrad_one_pass(C, C, D, P) && (h.s2a & 7) == 0;
    h.gbf = h.one_pass ? take((size_t)B * D * ((P + 31) / 32 * 32) * 2) : 0;
    h.total = off;
    return h;
}
static size_t head_weights_bytes(int32_t C, int32_t D) {
    return DgHeadWeightLayout(C, D).elems * 2;
}
extern "C" size_t dg_head_weights_bytes(int32_t C, int32_t D) {
    if (head_check(1, C, D, 1) != DG_OK) return 0;
    return head_weights_bytes(C, D);
}

// images Bs.. of a tensor that continues in a second allocation: (second base - first base) in elements, minus the Bs images in front
template <typename T>
static long long pair_delta(const T* first, const T* second, int32_t Bs, long long stride) {
    if (!second) return 0;
    return (long long)((reinterpret_cast<intptr_t>(second) - reinterpret_cast<intptr_t>(first)) / (intptr_t)sizeof(T)) - (long long)Bs * stride;
}

// B images in all; the first Bs from feat / into code / feats_out, the rest from / into the *2 tensors (null: one tensor, Bs = B)
static int head_forward_impl(int32_t B, int32_t Bs, int32_t C, int32_t D, int32_t P, const float* feat, const float* feat2,
                             const float* w1, const float* b1, const float* w2a, const float* b2a, const float* w2b, const float* b2b,
                             const float* keep1, const float* keep2, const float* keep3, float keep_scale,
                             float* code, float* code2, float* feats_out, float* feats_out2, void* hidden, void* wscratch, dg_stream_t stream_) {
    if (int rc = head_check(B, C, D, P)) return rc;
    if (!feat || !w1 || !b1 || !code || !wscratch) return fail(DG_ERR_INVALID, "null pointer");
    if (Bs < B && (!feat2 || !code2 || ((feats_out != nullptr) != (feats_out2 != nullptr)))) return fail(DG_ERR_INVALID, "null pointer of the second pass");
    const bool nonlinear = w2a != nullptr;
    if (nonlinear && (!b2a || !w2b || !b2b)) return fail(DG_ERR_INVALID, "cluster2 needs all four of its tensors");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    DG_HIP(dg_launch_head_prep(w1, w2a, w2b, wscratch, C, D, s));
    DgHeadFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = feat; a.w1 = w1; a.b1 = b1; a.w2a = w2a; a.b2a = b2a; a.w2b = w2b; a.b2b = b2b;
    { const DgHeadWeightLayout L(C, D); const __bf16* w = static_cast<const __bf16*>(wscratch); a.w1_bf = w + L.w1; a.w2a_bf = w + L.w2a; a.w2b_bf = w + L.w2b; }
    a.keep1 = keep1; a.keep2 = keep2; a.keep3 = keep3; a.scale = keep_scale;
    a.code = code; a.feats_out = feats_out; a.hidden = static_cast<__bf16*>(hidden);
    a.B = B; a.C = C; a.D = D; a.P = P;
    a.Bs = Bs;
    a.d_feat = pair_delta(feat, feat2, Bs, (long long)C * P);
    a.d_code = pair_delta(code, code2, Bs, (long long)D * P);
    a.d_fo = pair_delta(feats_out, feats_out2, Bs, (long long)C * P);
    DG_HIP(dg_launch_head_fwd(a, s));
    return DG_OK;
}

extern "C" int dg_head_forward(int32_t B, int32_t C, int32_t D, int32_t P, const float* feat,
                               const float* w1, const float* b1, const float* w2a, const float* b2a, const float* w2b, const float* b2b,
                               const float* keep1, const float* keep2, const float* keep3, float keep_scale,
                               float* code, float* feats_out, void* hidden, void* wscratch, dg_stream_t stream_) {
    return head_forward_impl(B, B, C, D, P, feat, nullptr, w1, b1, w2a, b2a, w2b, b2b, keep1, keep2, keep3, keep_scale, code, nullptr,
                             feats_out, nullptr, hidden, wscratch, stream_);
}

extern "C" int dg_head_forward_pair(int32_t B, int32_t C, int32_t D, int32_t P, const float* feat, const float* feat_pos,
                                    const float* w1, const float* b1, const float* w2a, const float* b2a, const float* w2b, const float* b2b,
                                    const float* keep1, const float* keep2, const float* keep3, float keep_scale,
                                    float* code, float* code_pos, float* feats_out, float* feats_out_pos, void* hidden, void* wscratch,
                                    dg_stream_t stream_) {
    if (B < 1 || B > (1 << 20)) return fail(DG_ERR_INVALID, "bad head dimensions");
    return head_forward_impl(2 * B, B, C, D, P, feat, feat_pos, w1, b1, w2a, b2a, w2b, b2b, keep1, keep2, keep3, keep_scale, code, code_pos,
                             feats_out, feats_out_pos, hidden, wscratch, stream_);
}

extern "C" int dg_normalize_split(int32_t B, int32_t C, int32_t h, int32_t w, const float* src, int32_t nchunks, int32_t chunk_c,
                                  float* const* dst, dg_stream_t stream_) {
    if (B < 1 || C < 1 || h < 1 || w < 1 || !src || !dst) return fail(DG_ERR_INVALID, "dg_normalize_split: bad arguments");
    if (nchunks < 1 || nchunks > 16 || chunk_c < 1 || (long long)chunk_c * (nchunks - 1) >= C || (long long)chunk_c * nchunks < C)
        return fail(DG_ERR_INVALID, "dg_normalize_split: %d chunks of %d channels do not tile C=%d (at most 16 chunks)", nchunks, chunk_c, C);
    for (int k = 0; k < nchunks; ++k) if (!dst[k]) return fail(DG_ERR_INVALID, "dg_normalize_split: null destination %d", k);
    DG_HIP(dg_launch_normalize_split(src, B, C, h * w, nchunks, chunk_c, dst, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" size_t dg_head_workspace_bytes(int32_t B, int32_t C, int32_t D, int32_t P) {
    if (head_check(B, C, D, P) != DG_OK) return 0;
    return head_plan(B, C, D, P).total;
}

static int head_backward_impl(int32_t B, int32_t Bs, int32_t C, int32_t D, int32_t P, const float* feat, const float* feat2, const float* keep1,
                              const float* keep2, float keep_scale, const void* hidden, const void* wscratch, const float* grad_code,
                              const float* grad_code2, float* grad_w1, float* grad_b1, float* grad_w2a, float* grad_b2a, float* grad_w2b,
                              float* grad_b2b, void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (int rc = head_check(B, C, D, P)) return rc;
    if (!feat || !grad_code || !grad_w1 || !grad_b1 || !workspace) return fail(DG_ERR_INVALID, "null pointer");
    if (Bs < B && (!feat2 || !grad_code2)) return fail(DG_ERR_INVALID, "null pointer of the second pass");
    const long long d_feat = pair_delta(feat, feat2, Bs, (long long)C * P), d_g = pair_delta(grad_code, grad_code2, Bs, (long long)D * P);
    const bool nonlinear = grad_w2a != nullptr;
    if (nonlinear && (!hidden || !wscratch || !grad_b2a || !grad_w2b || !grad_b2b)) return fail(DG_ERR_INVALID, "null cluster2 pointer");
    const HeadPlan h = head_plan(B, C, D, P);
    if (workspace_bytes < h.total) return fail(DG_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, h.total);
    hipStream_t s = static_cast<hipStream_t>(stream_);
    char* ws = static_cast<char*>(workspace);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    DgHeadReduceArgs red;
    memset(&red, 0, sizeof(red));
    auto reduce = [&](const float* part, float* out, float* out2, int n, int splits, float scale) {
        red.jobs[red.njobs++] = DgHeadReduceJob{part, out, out2, n, splits, scale};
    };
    // d W1[d][k] = scale * keep1[b][k] * sum_p g[d][p] f[k][p]      (with cluster2: in the launch of d W2a below, which reads the same f)
    if (!nonlinear) {
        DgHeadWgradArgs w{grad_code, feat, keep1, F32(h.p1), B, D, C, P, h.s1};
        w.A2 = nullptr; w.keep_2 = nullptr; w.part2 = nullptr; w.M2 = 0;
        w.Bs = Bs; w.dA = d_g; w.dB = d_feat; w.dA2 = 0;
        DG_HIP(dg_launch_head_wgrad(w, false, false, s));
        reduce(F32(h.p1), grad_w1, nullptr, D * C, h.s1, keep1 ? keep_scale : 1.f);
    }
    if (!nonlinear) {        // d b1 = row sums of d code
        DG_HIP(dg_launch_head_rowsum(grad_code, false, grad_b1, nullptr, B, D, P, s, Bs, d_g));
        DG_HIP(dg_launch_head_reduce(red, s));
        return DG_OK;
    }
    __bf16* dh = reinterpret_cast<__bf16*>(ws + h.dh);
    const __bf16* w2bT = static_cast<const __bf16*>(wscratch) + DgHeadWeightLayout(C, D).w2bT;
    DgHeadDhArgs d{grad_code, w2bT, static_cast<const __bf16*>(hidden), dh, F32(h.pbd), F32(h.pb2a), B, C, D, P, Bs, d_g};
    d.gcode_bf = h.one_pass ? reinterpret_cast<__bf16*>(ws + h.gbf) : nullptr;
    d.step_major = (h.one_pass && h.dh_blocks > 0) ? 1 : 0;          // (both ends are this round's kernels: k_head_dh2 writes what k_head_wgrad3 reads)
    // d W2b = d code x hidden^T needs nothing of k_head_dh: it runs BESIDE it on the library's second stream where there is one (fork
    // / join by events, capturable): two launches that each leave most of the chip idle (27 and 37 us at the paired headline shape)
    DgHeadWgradArgs wb{grad_code, hidden, nullptr, F32(h.p2b), B, D, C, P, h.s2b};
    wb.A2 = nullptr; wb.keep_2 = nullptr; wb.part2 = nullptr; wb.M2 = 0;
    wb.Bs = Bs; wb.dA = d_g; wb.dB = 0; wb.dA2 = 0;
    const bool w2b_fused = h.dh_blocks > 0;              // (k_head_dh2: the product rides in the d hidden launch - no second launch, no second stream)
    if (w2b_fused) d.part_w2b = F32(h.p2b);
    std::optional<SideRegion> side_o;
    if (!w2b_fused) side_o.emplace(s);
    const bool side = side_o && *side_o;
    if (side) {
        DG_HIP(side_o->fork());
        DG_HIP(dg_launch_head_wgrad(wb, false, true, side_o->stream()));
        DG_HIP(side_o->record_join());
    }
    DG_HIP(dg_launch_head_dh(d, s));
    reduce(F32(h.pbd), grad_b1, grad_b2b, D, B * h.tiles, 1.f);       // d b1 = d b2b = row sums of d code
    reduce(F32(h.pb2a), grad_b2a, nullptr, C, B * h.tiles, 1.f);
    if (!side && !w2b_fused) DG_HIP(dg_launch_head_wgrad(wb, false, true, s));
    reduce(F32(h.p2b), grad_w2b, nullptr, D * C, h.s2b, 1.f);
    DgHeadWgradArgs wa{dh, feat, keep2, F32(h.p2a), B, C, C, P, h.s2a, grad_code, keep1, F32(h.p1), D};
    wa.Bs = Bs; wa.dA = 0; wa.dB = d_feat; wa.dA2 = d_g;
    wa.A2h = d.gcode_bf;
    wa.a_step_major = d.step_major;
    DG_HIP(dg_launch_head_wgrad(wa, true, false, s));
    reduce(F32(h.p2a), grad_w2a, nullptr, C * C, h.s2a, keep2 ? keep_scale : 1.f);
    reduce(F32(h.p1), grad_w1, nullptr, D * C, h.s2a, keep1 ? keep_scale : 1.f);
    if (side) DG_HIP(side_o->join());
    DG_HIP(dg_launch_head_reduce(red, s));         // all five reductions in one launch
    return DG_OK;
}

extern "C" int dg_head_backward(int32_t B, int32_t C, int32_t D, int32_t P, const float* feat, const float* keep1, const float* keep2,
                                float keep_scale, const void* hidden, const void* wscratch, const float* grad_code,
                                float* grad_w1, float* grad_b1, float* grad_w2a, float* grad_b2a, float* grad_w2b, float* grad_b2b,
                                void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    return head_backward_impl(B, B, C, D, P, feat, nullptr, keep1, keep2, keep_scale, hidden, wscratch, grad_code, nullptr, grad_w1, grad_b1,
                              grad_w2a, grad_b2a, grad_w2b, grad_b2b, workspace, workspace_bytes, stream_);
}

extern "C" int dg_head_backward_pair(int32_t B, int32_t C, int32_t D, int32_t P, const float* feat, const float* feat_pos, const float* keep1,
                                     const float* keep2, float keep_scale, const void* hidden, const void* wscratch, const float* grad_code,
                                     const float* grad_code_pos, float* grad_w1, float* grad_b1, float* grad_w2a, float* grad_b2a,
                                     float* grad_w2b, float* grad_b2b, void* workspace, size_t workspace_bytes, dg_stream_t stream_) {
    if (B < 1 || B > (1 << 20)) return fail(DG_ERR_INVALID, "bad head dimensions");
    return head_backward_impl(2 * B, B, C, D, P, feat, feat_pos, keep1, keep2, keep_scale, hidden, wscratch, grad_code, grad_code_pos, grad_w1,
                              grad_b1, grad_w2a, grad_b2a, grad_w2b, grad_b2b, workspace, workspace_bytes, stream_);
}

extern "C" int dg_cluster_lookup_forward(const float* x, const float* clusters, float alpha, int32_t B, int32_t D, int32_t n, int32_t P,
                                         float* inner, float* probs, float* logp, float* loss, float* scratch, dg_stream_t stream_) {
    if (B < 1 || D < 1 || n < 1 || P < 1) return fail(DG_ERR_INVALID, "bad cluster-lookup dimensions");
    if (D > 128 || (size_t)n * (D + 1) > 16000) return fail(DG_ERR_UNSUPPORTED, "cluster lookup needs D <= 128 and n * (D + 1) <= 16000");
    if (!x || !clusters || !inner || !loss || !scratch) return fail(DG_ERR_INVALID, "null pointer");
    DgClusterArgs a{x, clusters, alpha, inner, probs, logp, scratch, B, D, n, P};
    DG_HIP(dg_launch_cluster_fwd(a, loss, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

extern "C" int dg_cluster_lookup_backward(const float* x, const float* clusters, const float* inner, float alpha, const float* grad_loss,
                                          int32_t B, int32_t D, int32_t n, int32_t P, float* grad_clusters, float* grad_x, float* scratch,
                                          dg_stream_t stream_) {
    if (B < 1 || D < 1 || n < 1 || P < 1) return fail(DG_ERR_INVALID, "bad cluster-lookup dimensions");
    if (D > 128 || (size_t)n * (D + 1) > 16000 || (size_t)(n + D) * 65 * 4 > 160 * 1024) return fail(DG_ERR_UNSUPPORTED, "cluster lookup needs D <= 128 and n * (D + 1) <= 16000");
    if (!x || !clusters || !inner || !grad_loss || !grad_clusters || !scratch) return fail(DG_ERR_INVALID, "null pointer");
    DgClusterBwdArgs a{x, clusters, inner, grad_loss, alpha, scratch, grad_x, scratch + (size_t)B * n * P, grad_clusters, B, D, n, P};
    DG_HIP(dg_launch_cluster_bwd(a, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

static int probe_check(int32_t B, int32_t n, int32_t h, int32_t w, int32_t H, int32_t W) {
    if (B < 1 || n < 1 || h < 1 || w < 1 || H < 1 || W < 1) return fail(DG_ERR_INVALID, "bad probe dimensions");
    if (n * w > 2048 || (size_t)(2 * n * w + (size_t)n * W) * 4 > 150 * 1024) return fail(DG_ERR_UNSUPPORTED, "probe loss needs n*w <= 2048 and n*(2w+W) floats of LDS");
    return DG_OK;
}
extern "C" int dg_probe_ce_forward(const float* logits, const int64_t* label, int32_t B, int32_t n, int32_t h, int32_t w, int32_t H,
                                   int32_t W, float* out3, float* scratch, dg_stream_t stream_) {
    if (int rc = probe_check(B, n, h, w, H, W)) return rc;
    if (!logits || !label || !out3 || !scratch) return fail(DG_ERR_INVALID, "null pointer");
    DgProbeCeArgs a{logits, label, scratch, nullptr, nullptr, nullptr, B, n, h, w, H, W};
    DG_HIP(dg_launch_probe_ce_fwd(a, out3, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}
extern "C" int dg_probe_ce_backward(const float* logits, const int64_t* label, const float* out3, const float* grad_loss, int32_t B,
                                    int32_t n, int32_t h, int32_t w, int32_t H, int32_t W, float* grad_logits, dg_stream_t stream_) {
    if (int rc = probe_check(B, n, h, w, H, W)) return rc;
    if (!logits || !label || !out3 || !grad_loss || !grad_logits) return fail(DG_ERR_INVALID, "null pointer");
    DgProbeCeArgs a{logits, label, nullptr, grad_loss, out3, grad_logits, B, n, h, w, H, W};
    DG_HIP(dg_launch_probe_ce_bwd(a, static_cast<hipStream_t>(stream_)));
    return DG_OK;
}

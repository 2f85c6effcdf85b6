/*
 * depthg_corr.h - C ABI of the MI355X (gfx950) implementation of DepthG's feature-correlation
 * loss hot path.  Plain C: pointers and sizes only, no torch types.
 *
 * The reference (leonsick/depthg) has no FFI; the boundary it offers for this path is the
 * Python nn.Module `ContrastiveCorrelationLoss` (src/modules.py:1221-1367).  The host-side
 * mirror of that module lives in depthg_amd/loss.py and binds these entry points with ctypes
 * (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless stated otherwise;
 *  - the callee never allocates or frees: scratch comes from a caller-sized workspace whose
 *    size is returned by dg_corr_workspace_bytes();
 *  - every launch goes to the given hipStream_t and is asynchronous w.r.t. the host;
 *  - return value 0 = ok, negative = error (dg_last_error() gives text); no exceptions.
 *  - one host thread per device; state kept across calls: the thread-local last-error string and a per-kernel cache of the
 *    dynamic-LDS attribute.
 */
#ifndef DEPTHG_CORR_H
#define DEPTHG_CORR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_VERSION 118   /* 118: dg_sampled_sumsq, dg_corr_forward_extnorm (feature maps wider than 768 channels on SAMPLED grids above 160 positions, in channel chunks); 117: DG_FEATS_UNIT, dg_normalize_split (feature maps wider than 768 channels on the dense identity grid, in chunks of the width the operand kernels hold: the loss is linear in the feature correlation); 116: dg_prof_main_span takes FOUR words (+ the workgroups' lifetimes in shader cycles and wall ticks: the clock the CUs held); 115: dg_corr_intra_folded; 114: dg_fps_coords_pair takes a workspace (dg_fps_workspace_bytes(2 B, h, w): the pooled depth maps, written by a launch over the whole chip in front of the sampler), dg_corr_materialize_shared, dg_prof_main_span (the fused correlation launch's execution span inside a replayed step), sample grids of <= 160 positions at any feature width (fused small-grid kernel); 113: dg_corr_forward_masked; 112: dg_rand_coords_state; 111: dg_head_forward_pair / dg_head_backward_pair; 110: dg_fps_coords_pair; 109: dg_knn_similarities; 108: dg_corr_desc.code_h / code_w (code maps of another resolution than the feature maps: the FeaturePyramidNet producer, src/modules.py:732-766), dg_corr_desc.flags DG_EXACT_MASKS; 107: dg_head_*, dg_cluster_lookup_*, dg_probe_ce_*; 106: dg_corr_main_kernel_name; 105: dg_corr_forward_draw; 104: dg_lhp_map_forward / dg_lhp_map_backward; 103: dg_super_perms_state; 102: DG_LINE_GRID, dg_salience_coords, dg_simple_depth_coords; 101: total weights, DG_OUT_TOTAL */

/* flags of dg_corr_desc.flags (names follow the cfg keys read at src/modules.py:1236-1352) */
#define DG_POINTWISE      (1u << 0)  /* cfg.pointwise: spatial centering of fd (modules.py:1236-1239) */
#define DG_ZERO_CLAMP     (1u << 1)  /* cfg.zero_clamp: clamp min 0 instead of -9999 (modules.py:1243-1246) */
#define DG_STABALIZE      (1u << 2)  /* cfg.stabalize: clamp max 0.8 (modules.py:1249-1250) */
#define DG_DEPTH_TERM     (1u << 3)  /* cfg.depth_feat_correlation_loss (modules.py:1334-1336) */
#define DG_NEED_GRAD      (1u << 4)  /* also produce d/d orig_code, d/d orig_code_pos pieces */
#define DG_SHARED_COORDS  (1u << 5)  /* coords1[n] == coords2[m] for all n, m (dense grid): negatives reuse
                                        the prepared operand of `feats`/`code` through the batch permutation */

#define DG_IDENTITY_GRID  (1u << 6)  /* with DG_SHARED_COORDS and S == h == w: both coords are the pixel-centre grid
                                        (coords[b,u,v] = (lin[v], lin[u]), lin = linspace(-1,1,S)): sample() is an exact
                                        transpose and the feats operands are built without bilinear taps */

#define DG_LINE_GRID      (1u << 7)  /* the sample grid is S x 1 instead of S x S: coords are (B,S,1,2), P = S positions and the
                                        un-reduced tensors are (B,1,S,1,S) - what depth_sampling='simple' feeds the loss
                                        (simple_depth_informed_sampling returns (B,n,1,2), src/modules.py:828-883,1299-1302);
                                        the depth term resizes depth to (1,S) (modules.py:1261-1262 with c1.shape[2:] = (1,S)) */

#define DG_EXACT_MASKS    (1u << 8)  /* build-side cfg key `dg_exact_masks`: on a gradient pass of the zero_clamp recipe the clamp mask
                                        1[cd >= 0] (modules.py:1250-1252) is decided by an fp32 cd wherever the fp16 cd the MFMAs
                                        compute is too close to zero to be trusted; without the flag the fp16 cd decides on the
                                        dense grids (the small sample grids always take exact masks) */

#define DG_FEATS_UNIT     (1u << 9)  /* with DG_IDENTITY_GRID: the channel vectors of orig_feats / orig_feats_pos are used as they are, not
                                        normalised - they are unit vectors already, or ONE CHANNEL CHUNK of unit vectors
                                        (dg_normalize_split).  The loss and its gradients are linear in the feature correlation
                                        fd (src/modules.py:797-809, 1231-1254: centering, shift and clamp(cd) * (fd - shift)), and fd is
                                        a sum over channels: a map of C > 768 channels is evaluated as one call per chunk - the first
                                        with the recipe's shifts and depth term, the others with zero shifts and no depth term - and
                                        the loss means (not the cd means) and code gradients add up (depthg_amd/loss.py does that). */
#define DG_MAX_NEG 8

/* error codes */
#define DG_OK               0
#define DG_ERR_INVALID     -1
#define DG_ERR_UNSUPPORTED -2
#define DG_ERR_WORKSPACE   -3
#define DG_ERR_LAUNCH      -4

typedef void* dg_stream_t; /* hipStream_t */

/* Shape/cfg descriptor of one loss call (one `ContrastiveCorrelationLoss.forward`). */
typedef struct dg_corr_desc {
    int32_t B;        /* batch (per rank) */
    int32_t C;        /* feature channels of orig_feats (384 ViT-S, 768 ViT-B); <= 768 per call on grids above 160 positions and the identity grid
                         (wider maps: one call per channel chunk - DG_FEATS_UNIT on the identity grid, dg_corr_forward_extnorm on sampled
                         coordinates), <= 8192 on smaller grids */
    int32_t D;        /* code channels = cfg.dim; <= 128 */
    int32_t h, w;     /* feature-map size of orig_feats / orig_feats_pos (and of the code maps unless code_h / code_w say otherwise) */
    int32_t S;        /* cfg.feature_samples; P = S*S positions are correlated (P = S with DG_LINE_GRID) */
    int32_t n_neg;    /* cfg.neg_samples (<= DG_MAX_NEG) */
    int32_t depth_h, depth_w; /* size of the depth map (image resolution); 0 if no depth */
    uint32_t flags;
    float shift_intra, shift_inter, shift_neg, shift_depth; /* cfg.pos_intra_shift ... cfg.depth_feat_shift */
    /* weights of the four loss means in the caller's total (src/train_segmentation.py:330-349: cfg.pos_intra_weight,
       pos_inter_weight, neg_inter_weight, depth_feat_weight, each times correspondence_weight - balance);
       out_scalars[DG_OUT_TOTAL] = their weighted sum.  All zero: no total wanted. */
    float w_intra, w_inter, w_neg, w_depth;
    /* size of orig_code / orig_code_pos when it differs from (h, w); 0, 0 = the same maps.  The reference's sample()
       (src/modules.py:822-825) takes normalised coordinates, so the loss accepts producers whose code map has another resolution
       than their feature map: FeaturePyramidNet returns low_res_feats (B,2048,7,7) next to code (B,dim,56,56), src/modules.py:732-766
       (its 2048 feature channels exceed this library's C <= 768).  Needs general coordinates (no DG_IDENTITY_GRID). */
    int32_t code_h, code_w;
} dg_corr_desc;

/* indices into out_scalars[] of dg_corr_forward */
enum {
    DG_OUT_LOSS_INTRA = 0, /* pos_intra_loss.mean()           (tuple element 0) */
    DG_OUT_LOSS_INTER = 1, /* pos_inter_loss.mean()           (element 2) */
    DG_OUT_LOSS_NEG   = 2, /* neg_inter_loss.mean()           (mean of element 4) */
    DG_OUT_LOSS_DEPTH = 3, /* depth_feat_loss.mean()          (element 6) */
    DG_OUT_CD_INTRA   = 4, /* pos_intra_cd.mean()             (mean of element 1) */
    DG_OUT_CD_INTER   = 5, /* pos_inter_cd.mean()             (mean of element 3) */
    DG_OUT_CD_NEG     = 6, /* neg_inter_cd.mean()             (mean of element 5) */
    DG_OUT_DD         = 7, /* depth_feat_cd.mean() (= mean of dd, element 7) */
    DG_OUT_TOTAL      = 8, /* w_intra*[0] + w_inter*[1] + w_neg*[2] + w_depth*[3]: the term training_step adds to its loss */
    DG_OUT_COUNT      = 9
};

int dg_version(void);
const char* dg_last_error(void);

/* Bytes of workspace dg_corr_forward/backward/materialize need for this descriptor. */
size_t dg_corr_workspace_bytes(const dg_corr_desc* desc);

/*
 * Forward of the whole loss (replaces ContrastiveCorrelationLoss.forward after the coordinate
 * draw, src/modules.py:1323-1367): sample() of feats/code at coords1, feats_pos/code_pos and the
 * n_neg negatives at coords2 (modules.py:822-825), norm() (modules.py:789-790), the
 * feature/code correlations (modules.py:797-809), pointwise centering, clamp, shift and the
 * mean reductions of helper() (modules.py:1231-1254) and of depth_feature_correlation()
 * (modules.py:1256-1278).
 *
 *  orig_feats, orig_feats_pos : fp32 (B,C,h,w) contiguous NCHW
 *  orig_code,  orig_code_pos  : fp32 (B,D,h,w), or (B,D,code_h,code_w) when the descriptor names a code-map size
 *  depth                      : fp32 (B,1,depth_h,depth_w) or NULL (required with DG_DEPTH_TERM)
 *  coords1, coords2           : fp32 (B,S,S,2) in [-1,1]  (what the reference passes to sample()); (B,S,1,2) with DG_LINE_GRID
 *  perms                      : int64 (n_neg,B) = super_perm() draws (modules.py:1184-1188,1341)
 *  out_scalars                : fp32 [DG_OUT_COUNT]
 * With DG_NEED_GRAD the unit-upstream gradient pieces are left in the workspace for
 * dg_corr_backward, which must be called with the SAME desc, coords, perms and workspace.
 */
int dg_corr_forward(const dg_corr_desc* desc,
                    const float* orig_feats, const float* orig_feats_pos,
                    const float* orig_code, const float* orig_code_pos,
                    const float* depth,
                    const float* coords1, const float* coords2, const int64_t* perms,
                    float* out_scalars,
                    void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * dg_corr_forward that also DRAWS the negatives' batch maps - the `super_perm` calls at the top of the reference's negative
 * loop (src/modules.py:1184-1188, 1340-1342) happen inside the forward there too.  perms_out (n_neg, B) is written (the same
 * values dg_super_perms_seeded(seed) / dg_super_perms_state(perm_state) would write: perm_state non-NULL selects the
 * device-resident generator and advances it, else `seed`) and must be handed to dg_corr_backward / dg_corr_materialize.  On
 * the identity grid the draw rides in the first launch of the forward (no launch of its own); with general coordinates, whose
 * first launch already reads the maps, it is launched first.
 */
int dg_corr_forward_draw(const dg_corr_desc* desc,
                         const float* orig_feats, const float* orig_feats_pos,
                         const float* orig_code, const float* orig_code_pos,
                         const float* depth,
                         const float* coords1, const float* coords2, int64_t* perms_out, uint64_t seed, void* perm_state,
                         float* out_scalars,
                         void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * dg_corr_forward / dg_corr_forward_draw with the Dropout2d of the two feature maps applied INSIDE the operand preparation
 * instead of by their producer (`feats = self.dropout(image_feat)`, nn.Dropout2d, src/modules.py:122-137: the last thing
 * DinoFeaturizer.forward does to the tensor this loss receives as orig_feats / orig_feats_pos).  The caller hands in the
 * UN-dropped features and the draw:
 *  feat_keep, feat_pos_keep : fp32 (B,C) keep flags 1 / 0 of orig_feats / orig_feats_pos, either may be NULL (no dropout)
 *  keep_scale               : 1/(1-p), applied to the kept channels
 * The operands are built from x * (keep * keep_scale) - the fp32 product the producer would have stored - so every output has
 * the bits of the call on the dropped features; what is saved is that tensor's round trip through HBM (38.5 MB written and read
 * per map at the headline shape).  Identity grid only (DG_IDENTITY_GRID); DG_ERR_UNSUPPORTED otherwise.
 *  perms, draw_perms, seed, perm_state : draw_perms != 0: as perms_out / seed / perm_state of dg_corr_forward_draw; 0: perms
 *                                        is read, as by dg_corr_forward.
 * dg_corr_backward does not read the feature maps.  (version 113)
 */
int dg_corr_forward_masked(const dg_corr_desc* desc,
                           const float* orig_feats, const float* orig_feats_pos,
                           const float* orig_code, const float* orig_code_pos,
                           const float* depth,
                           const float* coords1, const float* coords2, int64_t* perms, int32_t draw_perms, uint64_t seed,
                           void* perm_state, const float* feat_keep, const float* feat_pos_keep, float keep_scale,
                           float* out_scalars,
                           void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * Backward (replaces autograd through helper()/sample(), SURVEY.md section 9 "Gradient"):
 *  grad_scalars : fp32 [DG_OUT_COUNT] device = upstream gradient of the out_scalars vector: entries 0..3 (the four
 *                 loss means) and DG_OUT_TOTAL are used, effective d/d(loss mean i) = g[i] + g[DG_OUT_TOTAL] * w_i;
 *                 the cd means carry no gradient
 *                 (DG_OUT_LOSS_INTRA..DG_OUT_LOSS_DEPTH order)
 *  grad_code, grad_code_pos : fp32, the shape of orig_code ((B,D,h,w) or (B,D,code_h,code_w)), overwritten.
 *  coords1, coords2, perms  : the SAME arrays the forward of this workspace ran on.  With general coordinates the adjoint of
 *                 sample() gathers through inverse tap records that the FORWARD built from its coords (first launch of a
 *                 DG_NEED_GRAD forward); the backward does not rebuild them, so coords that differ from the forward's are
 *                 not detected and give the gradient at the forward's sample positions.  Only valid after a forward with
 *                 DG_NEED_GRAD on this workspace, before the next forward overwrites it.
 */
int dg_corr_backward(const dg_corr_desc* desc,
                     const float* grad_scalars,
                     const float* coords1, const float* coords2, const int64_t* perms,
                     float* grad_code, float* grad_code_pos,
                     void* workspace, size_t workspace_bytes, dg_stream_t stream);
/* Same with the upstream gradient of out_scalars[DG_OUT_TOTAL] alone (a device scalar): effective d/d(loss mean i) =
 * grad_total[0] * w_i.  What `total.backward()` needs - no 9-element gradient vector has to be assembled first. */
int dg_corr_backward_total(const dg_corr_desc* desc,
                           const float* grad_total,
                           const float* coords1, const float* coords2, const int64_t* perms,
                           float* grad_code, float* grad_code_pos,
                           void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * Optional full tensors the reference returns un-reduced (src/modules.py:1352-1367), computed
 * from the operands dg_corr_forward left in the workspace.  which: 0 = pos_intra, 1 = pos_inter,
 * 2+k = negative k, -1 = depth term.  out_cd / out_loss: fp32 (B,S,S,S,S) ((B,1,S,1,S) with DG_LINE_GRID) or NULL.
 * For which == -1 out_cd receives dd (tuple element 7).
 */
int dg_corr_materialize(const dg_corr_desc* desc, int32_t which,
                        float* out_cd, float* out_loss,
                        void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * dg_corr_materialize for calls with DG_SHARED_COORDS (the dense identity grid), where the negatives' operands are the anchors'
 * operand read through their batch maps: `perms` = the (n_neg, B) maps the forward used (dg_corr_forward's argument or what
 * dg_corr_forward_draw wrote).  The reference returns these tensors on every grid (src/modules.py:1352-1367; `hist_freq` steps of
 * src/train_segmentation.py histogram them).  Without DG_SHARED_COORDS it is dg_corr_materialize (perms may be NULL).
 */
int dg_corr_materialize_shared(const dg_corr_desc* desc, int32_t which, const int64_t* perms, float* out_cd, float* out_loss,
                               void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * Depth-guided sample locations (replaces farthest_point_sampling_depth, src/modules.py:999-1037,
 * with depth2points :988-996 and fps :939-985): adaptive_avg_pool2d of depth to (h,w), pinhole
 * back-projection with fov=90 (radians, quirk Q5), farthest point sampling of S*S points starting
 * at index 0, re-emitted in row-major order.
 *  depth      : fp32 (B,1,depth_h,depth_w)
 *  out_coords : fp32 (B,S,S,2), already mapped *2-1 as the caller does (modules.py:1305-1308)
 *  out_inds   : int32 (B,S*S) selection order (may be NULL)
 *  workspace  : >= dg_fps_workspace_bytes(B,h,w) bytes
 */
size_t dg_fps_workspace_bytes(int32_t B, int32_t h, int32_t w);
int dg_fps_coords(const float* depth, int32_t B, int32_t depth_h, int32_t depth_w,
                  int32_t h, int32_t w, int32_t S,
                  float* out_coords, int32_t* out_inds,
                  void* workspace, size_t workspace_bytes, dg_stream_t stream);
/* The two calls of one step (src/modules.py:1304-1308: the anchors' depth, then the positives') as one launch: the sampler
 * is sequential per image, so both run side by side on twice as many CUs and the caller needs no concatenated copy.
 * depth, depth_pos : fp32 (B,1,depth_h,depth_w) each;  out_coords (2B,S,S,2): [0,B) from depth, [B,2B) from depth_pos;
 * out_inds int32 (2B,S*S) or NULL.  (version 110) */
int dg_fps_coords_pair(const float* depth, const float* depth_pos, int32_t B, int32_t depth_h, int32_t depth_w,
                       int32_t h, int32_t w, int32_t S, float* out_coords, int32_t* out_inds,
                       void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * Salience-guided sample locations (replaces sample_nonzero_locations, src/modules.py:1191-1204, the cfg.use_salience
 * branch of the coordinate draw, :1290-1297): every position gets a uniformly drawn non-zero pixel of its image's
 * salience map (torch.nonzero = row-major order), returned as the reference does: both pixel coordinates divided by
 * the map HEIGHT (t.shape[1]), mapped *2-1 and flipped to (x, y).  The randomness comes from the caller as iid uniforms
 * in [0,1) (torch.rand); rank = min(int(u * count), count - 1).  An image without non-zeros gets
 * min(int(u_fallback * H), H - 1) for both coordinates (modules.py:1198).
 *  salience   : fp32 (B,H,W)         u_sel : fp32 (B,n)       u_fallback : fp32 (B,n,2)
 *  out_coords : fp32 (B,n,2); n = S*S, so the caller views it as (B,S,S,2)
 */
int dg_salience_coords(const float* salience, int32_t B, int32_t H, int32_t W, int32_t n,
                       const float* u_sel, const float* u_fallback, float* out_coords, dg_stream_t stream);

/*
 * Depth-distribution sample locations (replaces simple_depth_informed_sampling, src/modules.py:828-883, the
 * cfg.depth_sampling == 'simple' branch, :1299-1302): adaptive_max_pool2d of depth to (h,w), rounded to one decimal;
 * a depth value is drawn with probability count/h*w (the reference's multinomial over torch.unique counts), then one of
 * the pixels holding that value uniformly (torch.nonzero order).  u_value / u_pick are the caller's iid uniforms for
 * the two draws: sorted position R = min(int(u_value * h*w), h*w - 1) selects the value, min(int(u_pick * run), run - 1)
 * the pixel inside its run.
 *  depth      : fp32 (B,1,depth_h,depth_w)      u_value, u_pick : fp32 (B,n)      h*w <= 4096
 *  out_coords : fp32 (B,n,1,2) = ((row + .5) / h, (col + .5) / w) * 2 - 1 (the *2-1 of modules.py:1300 included);
 *               feed it to dg_corr_forward with DG_LINE_GRID and S = n.
 */
int dg_simple_depth_coords(const float* depth, int32_t B, int32_t depth_h, int32_t depth_w, int32_t h, int32_t w,
                           int32_t n, const float* u_value, const float* u_pick, float* out_coords, dg_stream_t stream);

/*
 * Confusion-matrix accumulation of the validation metrics (replaces the bincount of UnsupervisedMetrics.update /
 * update_cherry, src/utils.py:222-232, 279-289): stats[pred][actual] += 1 for every element with
 * 0 <= actual < n_classes and 0 <= pred < n_classes (the reference masks preds with n_classes as well, so the extra
 * clusters' rows stay zero).  Counts are exact integers: bit-identical to torch.bincount; under data parallelism the
 * matrices of the ranks are summed with one all-reduce (torchmetrics dist_reduce_fx="sum", src/utils.py:214-220).
 *  preds, target : int64 [count] (any shape, flattened)     stats : int64 (n_classes + extra_clusters, n_classes), in/out
 */
int dg_confusion_update(const int64_t* preds, const int64_t* target, int64_t count,
                        int32_t n_classes, int32_t extra_clusters, int64_t* stats, dg_stream_t stream);

/*
 * Depth propagation of the LHP branch (replaces LocalHiddenPositiveProjection.forward_depth up to its projection head,
 * src/modules.py:273-335: adaptive_avg_pool2d + depth2points(fov=90) per image, pairwise point distances, row-wise min-max
 * normalisation, map = 1 - d where d <= the row's 1 % quantile (torch.quantile, linear), out[:, p] = mean_q map[p][q] code[:, q]).
 * The (B,P,P) distance / map tensors are never formed.  Distances are the direct float32 formula; the reference's
 * torch.cdist takes its matmul path for P > 25, so outputs agree with it to about 1e-4 relative, not bit for bit.
 *  code   : fp32 (B,D,h,w), D <= 128, h*w <= 4096        depth : fp32 (B,1,depth_h,depth_w)
 *  out    : fp32 (B,D,h,w) = code_mixed
 *  points : fp32 (B,3,h*w) scratch, stats : fp32 (B,h*w,3) = per-row (min, max, quantile): both are inputs of the backward
 * dg_lhp_backward: grad_code[:, q] = (1/P) sum_p map[p][q] grad_out[:, p], with the same points / stats.
 */
int dg_lhp_forward(const float* code, const float* depth, int32_t B, int32_t D, int32_t h, int32_t w,
                   int32_t depth_h, int32_t depth_w, float* out, float* points, float* stats, dg_stream_t stream);
int dg_lhp_backward(const float* grad_out, const float* points, const float* stats, int32_t B, int32_t D, int32_t h, int32_t w,
                    float* grad_code, dg_stream_t stream);

/*
 * The other propagation maps of the LHP branch, up to the projection head.  `mode`:
 *  DG_LHP_ATTN        LocalHiddenPositiveProjection.forward_attn (src/modules.py:235-271): attn[:, :, 1:, 1:] averaged over
 *                     the heads, row-wise min-max normalised, zeroed above the row's 99 % quantile (torch.quantile, linear);
 *                     out[:, p] = mean_q map[p][q] code[:, q]
 *  DG_LHP_ORIG_DEPTH  OriginalLocalHiddenPositiveProjection.forward_depth (:436-487): map = 1 - normalised point distance,
 *                     zeroed where the distance is above the row mean, times the clipped 3x3 neighbourhood mask (:356-383);
 *                     out[:, p] = sum_q map[p][q] code[:, q] / divide_num[p]
 *  DG_LHP_ORIG_ATTN   OriginalLocalHiddenPositiveProjection.forward_attn (:403-434): heads-mean attention normalised with
 *                     the row's 10 % / 90 % quantiles, zeroed below the row mean, same mask and divisor
 * The reference leaves divide_num all zero (it is re-created inside the constructor loop and never filled, :354,382), so its
 * Original variants return sum / 0; `divide` is an input here so that both that behaviour and a repaired table can be run.
 *  code  : fp32 (B,D,h,w), D <= 128, h*w <= 4096      attn : fp32 (B,heads,h*w+1,h*w+1) (CLS row / column first)
 *  depth : fp32 (B,1,depth_h,depth_w) (ORIG_DEPTH)    divide : fp32 (h*w) (ORIG_*)
 *  out   : fp32 (B,D,h,w)
 *  map   : ATTN: fp32 (B,P,P), ORIG_*: fp32 (B,P,9): the weights, written by the forward and read by the backward
 *  points: fp32 (B,3,h*w) scratch (ORIG_DEPTH)
 * dg_lhp_map_backward: grad_code[:, q] = sum_p map[p][q] grad_out[:, p] * (1/P, or 1/divide[p]); no gradient reaches attn / depth
 * (the reference's come from a frozen backbone / the data loader).
 */
enum { DG_LHP_ATTN = 0, DG_LHP_ORIG_DEPTH = 1, DG_LHP_ORIG_ATTN = 2 };
int dg_lhp_map_forward(int32_t mode, const float* code, const float* attn, const float* depth, const float* divide,
                       int32_t B, int32_t D, int32_t h, int32_t w, int32_t heads, int32_t depth_h, int32_t depth_w,
                       float* out, float* map, float* points, dg_stream_t stream);
int dg_lhp_map_backward(int32_t mode, const float* grad_out, const float* map, const float* divide, int32_t B, int32_t D,
                        int32_t h, int32_t w, float* grad_code, dg_stream_t stream);

/*
 * The similarity slice of the offline nearest-neighbour search (replaces `pairwise_sims = torch.einsum("nf,mf->nm", batch_feats,
 * normed_feats)`, src/precompute_knns.py:106-108): out[i][j] = <queries[i], feats[j]>, fp32 products and fp32 accumulation on the
 * fp32 MFMA.
 *  queries : fp32 (rows_q, F), row stride q_stride elements      feats : fp32 (n, F), row stride f_stride      F >= 1
 *  out     : fp32 (rows_q, n), row stride out_stride
 * Any row stride >= F and any 4-byte-aligned pointer is accepted: rows are read with 16-byte loads when F is a multiple of 4 and
 * every row starts on a 16-byte boundary, with 4-byte loads otherwise - the same result bits either way.
 */
int dg_knn_similarities(const float* queries, const float* feats, int64_t rows_q, int64_t n, int32_t F, int64_t q_stride,
                        int64_t f_stride, float* out, int64_t out_stride, dg_stream_t stream);

/*
 * Row-wise top-k (replaces `torch.topk(pairwise_sims, 30)[1]` of the offline nearest-neighbour search,
 * src/precompute_knns.py:108-112; the similarity slice `einsum("nf,mf->nm")` itself is a plain GEMM and stays a library
 * call in the host mirror).  Row r of the result holds the column indices of the k largest values of row r, ordered by
 * value descending, ties by ascending column (torch.topk leaves the tie order unspecified).
 *  vals : fp32 (rows, cols) with row stride `row_stride` elements     k <= 64, k <= cols < 2^32
 *  out_idx : int64 (rows, k)        out_val : fp32 (rows, k) or NULL
 */
int dg_topk_rows(const float* vals, int64_t rows, int64_t cols, int64_t row_stride, int32_t k,
                 int64_t* out_idx, float* out_val, dg_stream_t stream);

/*
 * Negative-pair batch permutations (replaces super_perm, src/modules.py:1184-1188, called n_neg times per step at
 * :1336-1339): `count` independent uniform random permutations of 0..B-1 with fixed points bumped by one modulo B
 * (quirk Q6: B == 1 gives [0]).  The randomness comes from the caller: `keys` holds count*B iid uniforms (torch.rand);
 * row r of the result is the argsort of row r of the keys (ties by index), which is a uniform random permutation
 * like torch.randperm.
 *  keys : fp32 (count, B)      out : int64 (count, B)      B <= 8192
 */
int dg_super_perms(const float* keys, int32_t count, int32_t B, int64_t* out, dg_stream_t stream);
/* Same, with the keys drawn inside the launch: key(r, i) = Philox4x32-10(seed; counter r*B + i) >> 8, as a float in [0,1).
 * The caller supplies a fresh 64-bit seed per call (from its own RNG): one launch per step instead of rand + sort. */
int dg_super_perms_seeded(uint64_t seed, int32_t count, int32_t B, int64_t* out, dg_stream_t stream);
/* The same draws with the generator state ON THE DEVICE: state = {seed, draws so far, 0} (three 64-bit words owned by the
 * caller, zero the third); every call uses {seed, draws} as the Philox key and advances `draws` on the device.  Nothing
 * about the draw is baked into the launch, so a call recorded in a hipGraph (torch.cuda.graph around the training step)
 * yields new permutations on every replay - the reference draws them with the device generator at the same place,
 * src/modules.py:1184-1188,1336-1339. */
int dg_super_perms_state(uint64_t* state, int32_t count, int32_t B, int64_t* out, dg_stream_t stream);
/* The random sample coordinates of a step (`torch.rand(B, S, S, 2) * 2 - 1` for coords1 and coords2, src/modules.py:1310-1321) from
 * the same device-resident generator: n floats in [-1, 1) into `out`, one launch for both coordinate sets; advances the state.  For
 * steps recorded in a hipGraph (cfg.dg_graph_safe), where torch's generator costs two launches per tensor and two fills per replay;
 * the values are NOT torch's stream (neither are that mode's permutations).  (version 112) */
int dg_rand_coords_state(uint64_t* state, int64_t n, float* out, dg_stream_t stream);
/* Dropout2d keep flags of a graph-recorded step from the same generator: n floats, 1 with probability p_keep else 0 (what the head's
 * keep1 / keep2 / keep3 take; the reference draws them with nn.Dropout2d, src/modules.py:122-132).  One launch for all the masks of a
 * step; advances the state; not torch's stream.  (version 112) */
int dg_rand_keep_state(uint64_t* state, int64_t n, float p_keep, float* out, dg_stream_t stream);

/*
 * norm() of the reference (F.normalize(t, dim=1, eps=1e-10), src/modules.py:789-790) over ALL C channels of an NCHW map, written as
 * `nchunks` maps of `chunk_c` channels each (the last one: the rest), every one contiguous (B, c_k, h, w): the operands of
 * DG_FEATS_UNIT calls.  On the identity grid sample() is a transposition (reference quirk Q3), so normalising the map in front of it
 * is what the reference computes behind it.
 *   src (B,C,h,w) fp32; dst[k] (B, min(chunk_c, C - k chunk_c), h, w) fp32; 1 <= nchunks <= 16, chunk_c * (nchunks - 1) < C <= chunk_c * nchunks
 */
int dg_normalize_split(int32_t B, int32_t C, int32_t h, int32_t w, const float* src, int32_t nchunks, int32_t chunk_c,
                       float* const* dst, dg_stream_t stream);

/*
 * Feature maps wider than 768 channels on SAMPLED coordinates above 160 positions (the reference normalises behind sample(),
 * src/modules.py:789-790, 822-825: the norm of a sampled vector is over all of its channels).  Two calls per channel chunk:
 *   dg_sampled_sumsq: out[n][p] (+)= sum over the chunk's channels of sample(feats[srcidx ? srcidx[n] : n], coords[n])[c][p]^2
 *     feats (B,C_k,h,w) fp32 - ONE chunk, contiguous; coords (B,S,S,2) or (B,S,1,2) with line_grid; out (B,P) fp32; accumulate 0 / 1.
 *     Once per operand: orig_feats at coords1, orig_feats_pos at coords2 and - coordinates per image (no DG_SHARED_COORDS) - orig_feats
 *     through every negative's batch map at coords2 (src/modules.py:1341-1345).
 *   dg_corr_forward_extnorm: dg_corr_forward on the chunk with feat_inv = 1 / max(sqrt(sum over ALL chunks), 1e-10), (nops, B, P) fp32,
 *     operand-major (nops = 2, or 2 + n_neg without DG_SHARED_COORDS), used instead of the chunk's own norms.
 * As with DG_FEATS_UNIT the first chunk carries the recipe's shifts and depth term, the others zero shifts and none; loss means and code
 * gradients add up (dg_corr_backward* per chunk, unchanged).
 */
int dg_sampled_sumsq(int32_t B, int32_t C, int32_t h, int32_t w, int32_t S, int32_t line_grid, const float* feats,
                     const float* coords, const int64_t* srcidx, int32_t accumulate, float* out, dg_stream_t stream);
int dg_corr_forward_extnorm(const dg_corr_desc* desc, const float* orig_feats, const float* orig_feats_pos,
                            const float* orig_code, const float* orig_code_pos, const float* depth,
                            const float* coords1, const float* coords2, const int64_t* perms, const float* feat_inv,
                            float* out_scalars, void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * The segmentation head of DinoFeaturizer (replaces `cluster1(dropout(f)) + cluster2(dropout(f))` and the third
 * `dropout(f)` of src/modules.py:122-132, modules cluster1 / cluster2 of :75-88): 1x1 convolutions as bf16 MFMA products with
 * fp32 accumulation, Dropout2d and ReLU fused (a dropped channel is a zeroed weight column, 1/(1-p) scales the accumulator),
 * the fp32 features read once.
 *  feat                 : fp32 (B,C,P), P = h*w, C <= 768 and a multiple of 8
 *  w1,b1                : cluster1[0].weight (D,C), .bias (D); D <= 128
 *  w2a,b2a,w2b,b2b      : cluster2[0] (C,C),(C) and cluster2[2] (D,C),(D); all NULL for projection_type "linear"
 *  keep1,keep2,keep3    : fp32 (B,C) 1 = keep / 0 = drop: the three Dropout2d draws in the reference's order (cluster1's
 *                         input, cluster2's input, the returned feats); NULL = no dropout for that use (eval mode)
 *  keep_scale           : 1/(1-p)
 *  code                 : fp32 (B,D,P) out
 *  feats_out            : fp32 (B,C,P) out = feat * keep3 * keep_scale, or NULL
 *  hidden               : bf16 (B,C,P) out: cluster2's ReLU output, needed by dg_head_backward; may be NULL without one
 *  wscratch             : dg_head_weights_bytes(C, D) bytes: the forward leaves the bf16 copies of the weight matrices there
 *                         (its first launch); dg_head_backward reads them - keep the buffer until then
 */
size_t dg_head_weights_bytes(int32_t C, int32_t D);
int dg_head_forward(int32_t B, int32_t C, int32_t D, int32_t P, const float* feat,
                    const float* w1, const float* b1, const float* w2a, const float* b2a, const float* w2b, const float* b2b,
                    const float* keep1, const float* keep2, const float* keep3, float keep_scale,
                    float* code, float* feats_out, void* hidden, void* wscratch, dg_stream_t stream);
/* Bytes of scratch dg_head_backward needs (d hidden + the split partial sums of the three weight gradients). */
size_t dg_head_workspace_bytes(int32_t B, int32_t C, int32_t D, int32_t P);
/*
 * Gradients of the six head tensors from grad_code (B,D,P) (the backbone is frozen: nothing flows into feat).  Same feat /
 * keep1 / keep2 / keep_scale / hidden / wscratch as the forward.  grad_* : fp32, shapes of the parameters, overwritten; pass the
 * four cluster2 ones (and hidden) as NULL for projection_type "linear".  Bit-reproducible (no floating-point atomics).
 */
int dg_head_backward(int32_t B, int32_t C, int32_t D, int32_t P, const float* feat, const float* keep1, const float* keep2,
                     float keep_scale, const void* hidden, const void* wscratch, const float* grad_code,
                     float* grad_w1, float* grad_b1, float* grad_w2a, float* grad_b2a, float* grad_w2b, float* grad_b2b,
                     void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * Both featurizer passes of a training step in one set of launches (src/train_segmentation.py:303-306: `self.net(img)`, then
 * `self.net(img_pos)`, with the same weights): B images from feat, B from feat_pos, as if they were one batch of 2B - the weight
 * gradients come out summed over both passes (what autograd's accumulation of the two calls yields), one launch per kernel
 * instead of two, no concatenated copy of the features.  keep1/2/3, hidden: (2B, ...), the first pass's B rows first.
 * code / code_pos, feats_out / feats_out_pos, grad_code / grad_code_pos: separate (B, ...) tensors.  The backward's workspace:
 * dg_head_workspace_bytes(2 B, C, D, P).  (version 111)
 */
int dg_head_forward_pair(int32_t B, int32_t C, int32_t D, int32_t P, const float* feat, const float* feat_pos,
                         const float* w1, const float* b1, const float* w2a, const float* b2a, const float* w2b, const float* b2b,
                         const float* keep1, const float* keep2, const float* keep3, float keep_scale,
                         float* code, float* code_pos, float* feats_out, float* feats_out_pos, void* hidden, void* wscratch,
                         dg_stream_t stream);
int dg_head_backward_pair(int32_t B, int32_t C, int32_t D, int32_t P, const float* feat, const float* feat_pos, const float* keep1,
                          const float* keep2, float keep_scale, const void* hidden, const void* wscratch, const float* grad_code,
                          const float* grad_code_pos, float* grad_w1, float* grad_b1, float* grad_w2a, float* grad_b2a,
                          float* grad_w2b, float* grad_b2b, void* workspace, size_t workspace_bytes, dg_stream_t stream);

/*
 * ClusterLookup.forward (src/modules.py:664-675): inner = <normalize(x), normalize(clusters)>, probs = one-hot(arg-max) when
 * alpha is NaN (the reference's `alpha is None`) else softmax(alpha * inner), loss = -mean over (B,P) of sum_n probs * inner.
 *  x : fp32 (B,D,P)   clusters : fp32 (n,D), n * (D + 1) <= 16000   inner : fp32 (B,n,P) out (the backward reads it)
 *  probs, logp (= log_softmax(alpha * inner)) : fp32 (B,n,P) out or NULL     loss : fp32 [1] out
 *  scratch : fp32 [B * ceil(P/256)]
 * dg_cluster_lookup_backward: grad_clusters (n,D) and, when grad_x is not NULL, grad_x (B,D,P) for the upstream grad_loss [1]
 * (device).  scratch : fp32 [B*n*P + B*ceil(P/64)*n*D].
 */
int dg_cluster_lookup_forward(const float* x, const float* clusters, float alpha, int32_t B, int32_t D, int32_t n, int32_t P,
                              float* inner, float* probs, float* logp, float* loss, float* scratch, dg_stream_t stream);
int dg_cluster_lookup_backward(const float* x, const float* clusters, const float* inner, float alpha, const float* grad_loss,
                               int32_t B, int32_t D, int32_t n, int32_t P, float* grad_clusters, float* grad_x, float* scratch,
                               dg_stream_t stream);

/*
 * The linear probe's loss (src/train_segmentation.py:427-434): logits (B,n,h,w) resized to the label resolution with
 * F.interpolate(mode="bilinear", align_corners=False), cross entropy over the pixels with 0 <= label < n, mean.
 *  label : int64 (B,H,W)    out3 : fp32 [3] = {sum of -log p[label], labelled pixels, loss}    scratch : fp32 [2*B*H]
 * dg_probe_ce_backward: grad_logits (B,n,h,w) for the upstream grad_loss [1]; out3 as the forward wrote it.  n*w <= 2048.
 */
int dg_probe_ce_forward(const float* logits, const int64_t* label, int32_t B, int32_t n, int32_t h, int32_t w, int32_t H, int32_t W,
                        float* out3, float* scratch, dg_stream_t stream);
int dg_probe_ce_backward(const float* logits, const int64_t* label, const float* out3, const float* grad_loss, int32_t B, int32_t n,
                         int32_t h, int32_t w, int32_t H, int32_t W, float* grad_logits, dg_stream_t stream);

/* Measurement aid: name of the kernel the fused correlation launch of this descriptor runs ("k_corr2": the one-wave-per-SIMD
 * form of dg_corr2.hip, "k_corr_main": the general form), decided by the same predicate the launch uses; NULL on a bad desc. */
const char* dg_corr_main_kernel_name(const dg_corr_desc* desc);

/* Measurement aid: 1 when the fused correlation launch of this descriptor also forms the intra pair-set's streamed-side code
 * gradient (k_corr2's FOLD: helper(feats, feats, code, code) of src/modules.py:1236-1254 is symmetric up to its row-mean centering;
 * DESIGN.md section 4.1) - that share of the algorithmic work then belongs to the fused launch, not to k_gs; 0 otherwise; -1 on a
 * bad desc.  (DG_FOLD_INTRA=0 in the environment switches the fold off.) */
int dg_corr_intra_folded(const dg_corr_desc* desc);

/*
 * Measurement aid (bench.py roofline leg): the execution span of the fused correlation launch INSIDE the step, hipGraph replays
 * included, and the shader clock its CUs held.  `span` = device pointer to FOUR uint64 (or NULL: off).  Every workgroup of the fused
 * kernel of every later dg_corr_forward* call of this process takes min(span[0], entry time) and max(span[1], exit time) and adds its
 * own lifetime to span[2] (shader cycles, s_memtime) and span[3] (ticks of the GPU's constant 100-MHz wall clock, s_memrealtime:
 * 10 ns per tick), all with device-scope atomics.  The caller sets {UINT64_MAX, 0, 0, 0} in front of the step it wants to read (e.g.
 * a copy on the launch stream) and reads behind it: (span[1] - span[0]) x 10 ns = the interval a kernel trace reports for the
 * launch, minus the dispatch ramp; span[2] / span[3] x 0.1 = the clock in GHz the kernel's CUs held on average.  Nothing in the
 * product path depends on it.  LIFETIME: the pointer in force when a launch is RECORDED is what the launch uses - a hipGraph captured
 * while a span was set writes to it at every replay, so the four words must outlive every such graph (or the graph must be dropped
 * first); capture the graphs you time with the span off (NULL) and a separate one with it on for the measurement.
 */
int dg_prof_main_span(void* span);

/*
 * Measurement aid (bench.py roofline leg): re-launch only the fused correlation kernel on the operands a
 * previous dg_corr_forward with the same desc / perms / workspace prepared.  Idempotent.
 */
int dg_corr_relaunch_main(const dg_corr_desc* desc, const int64_t* perms,
                          void* workspace, size_t workspace_bytes, dg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEPTHG_CORR_H */

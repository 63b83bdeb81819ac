/*
 * gs_rasterizer.h -- C ABI of the MI355X-native differentiable Gaussian rasterizer
 * (libgs_rasterizer_hip.so, built from 4dgs-slam_amd/csrc/).
 *
 * This is the drop-in boundary for the reference's raw-pointer rasterizer core
 *   CudaRasterizer::Rasterizer::{forward,backward,markVisible}
 *   (submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:24-88,
 *    implemented in cuda_rasterizer/rasterizer_impl.cu:141-153,198-344,348-455)
 * as called by the torch glue in rasterize_points.cu:35-232. Plain pointers and sizes only:
 * every pointer is a DEVICE pointer into memory owned by the caller (PyTorch in practice),
 * fp32 / int32, contiguous, laid out exactly as the reference lays them out. A NULL pointer
 * selects the same branch a nullptr selects in the reference (shs / colors_precomp /
 * scales+rotations / cov3D_precomp).
 *
 * Differences from the reference core, all on the safe side:
 *   - the three std::function<char*(size_t)> resize callbacks (rasterizer.h:27-29,
 *     rasterize_points.cu:27-33) become plain C callbacks gsr_alloc_fn + user pointer;
 *   - an explicit hipStream_t (passed as void*) instead of the legacy default stream;
 *   - errors are returned (negative codes + gsr_last_error()) instead of thrown / __trap'd;
 *   - backward fully overwrites every gradient output, so the caller does not have to
 *     zero-fill them first (rasterize_points.cu:160-170 does; doing so stays harmless).
 * The byte layout inside the three scratch buffers is private to the library (as it is in the
 * reference); they must be handed back unchanged to gsr_backward together with R.
 */
#ifndef GS_RASTERIZER_H_INCLUDED
#define GS_RASTERIZER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_BLOCK_X 16 /* cuda_rasterizer/config.h:15-17 */
#define GSR_BLOCK_Y 16
#define GSR_NUM_CHANNELS 3

/* error codes (negative return values) */
#define GSR_ERR_INVALID_ARGUMENT (-1)
#define GSR_ERR_HIP (-2)          /* a HIP runtime call failed; text in gsr_last_error() */
#define GSR_ERR_PREFILTERED (-3)  /* auxiliary.h:156-160: a point was culled although prefiltered was set */
#define GSR_ERR_ALLOC (-4)        /* an allocation callback returned NULL */

/* Replaces std::function<char*(size_t)> (rasterizer.h:27-29): must return a device pointer to at
 * least `bytes` bytes (any alignment; the library aligns internally) that stays valid until the
 * matching gsr_backward call has been enqueued. */
typedef char* (*gsr_alloc_fn)(void* user, size_t bytes);

/* Rasterizer::forward (rasterizer.h:24-52 / rasterizer_impl.cu:198-344).
 * Returns num_rendered R >= 0 (the number of Gaussian x tile instances), or a negative error code.
 * out_color[3,H,W], out_depth[1,H,W], out_opacity[1,H,W], radii[P], n_touched[P] are fully written
 * (n_touched is zeroed by the library before accumulation). radii may be NULL (internal radii are used).
 * Performs exactly one host synchronisation on `stream` (to learn R), like rasterizer_impl.cu:284. */
int gsr_forward(gsr_alloc_fn geometry_alloc, void* geometry_user,
                gsr_alloc_fn binning_alloc, void* binning_user,
                gsr_alloc_fn image_alloc, void* image_user,
                int P, int D, int M,
                const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered,
                float* out_color, float* out_depth, float* out_opacity, int* radii, int* n_touched,
                int debug, void* stream);

/* Rasterizer::backward (rasterizer.h:54-88 / rasterizer_impl.cu:348-455). Returns 0 or a negative error code.
 * Outputs (all fully written): dL_dmean2D[P,3] (z = 0), dL_dconic[P,2,2] (element [1][0] = 0), dL_dopacity[P],
 * dL_dcolor[P,3], dL_ddepth[P], dL_dmean3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3] (may be NULL when M == 0 or shs == NULL),
 * dL_dscale[P,3], dL_drot[P,4] (zero when scales == NULL), dL_dtau[P,6] = [rho(3), theta(3)] per Gaussian.
 * Unlike the reference (float atomics, backward.cu:774-783) the result is bit-reproducible run to run. */
int gsr_backward(int P, int D, int M, int R,
                 const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* projmatrix_raw, const float* campos,
                 float tan_fovx, float tan_fovy, const int* radii,
                 char* geom_buffer, char* binning_buffer, char* image_buffer,
                 const float* dL_dpix, const float* dL_dpix_depth,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dtau,
                 int debug, void* stream);

/* gsr_backward plus one fused extra: dL_dtau_sum[6] = sum over Gaussians of dL_dtau (what the reference's Python layer computes
 * with torch.sum, DGR/diff_gaussian_rasterization/__init__.py:152-154). Pass NULL to skip.
 * Here -- not in gsr_backward -- gradients that only feed other gradients inside the kernel may be NULL and are then not
 * written: dL_dconic, dL_dcolor, dL_ddepth, dL_dcov3D, and dL_dtau when dL_dtau_sum is given (80 of the 148 bytes stored
 * per Gaussian). The reference allocates and fills all of them (rasterize_points.cu:160-170) although its autograd Function
 * drops three (DGR/diff_gaussian_rasterization/__init__.py:139-151).
 * `debug` carries one more bit here and in gsr_backward_raw: GSR_BACKWARD_ACCUMULATE. With it the PARAMETER gradients (dL_dmean3D,
 * dL_dsh, dL_dopacity, dL_dscale, dL_drot; raw mode: the gsr_raw_grads tensors of the model parameters, not those of the deltas) are
 * ADDED to what the buffers hold, for visible Gaussians only; rows of invisible Gaussians are not touched and nothing is zero-filled.
 * A caller that sums the views of one optimizer step into one gradient buffer (the mapping loop: 8-64 keyframes) zero-fills once per
 * step and thereby skips the zero rows this call would write per view (71 % of the Gaussians at BASELINE config #5) and the
 * read-modify-write of its own accumulation. dL_dmean2D (a per-view statistic) and the intermediate gradients are still overwritten. */
#define GSR_BACKWARD_ACCUMULATE 2
/* GSR_BACKWARD_POSE_ONLY (another bit of `debug`): the caller wants no parameter gradients at all (camera tracking reads only the pose
 * gradient dL_dtau_sum and dL_dmean2D): dL_dmean3D, dL_dsh, dL_dopacity, dL_dscale, dL_drot (raw mode: every gsr_raw_grads pointer) may be
 * NULL and are not written; the covariance -> scale / rotation chain is skipped. */
#define GSR_BACKWARD_POSE_ONLY 4
int gsr_backward_fused(int P, int D, int M, int R,
                       const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* projmatrix_raw, const float* campos,
                       float tan_fovx, float tan_fovy, const int* radii,
                       char* geom_buffer, char* binning_buffer, char* image_buffer,
                       const float* dL_dpix, const float* dL_dpix_depth,
                       float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                       float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dtau,
                       float* dL_dtau_sum, int debug, void* stream);

/* ---- fused prologue: the model's raw parameters in, their gradients out --------------------------
 * The reference's render() turns GaussianModel parameters into rasterizer inputs with ~10 elementwise torch kernels per view
 * (gaussian_splatting/gaussian_renderer/__init__.py:108-127,159-174; scene/gaussian_model.py:60-68,100-128) and autograd
 * runs as many again on the way back:
 *     means3D = _xyz + scatter(dx -> dygs);        scales = exp(_scaling) [.repeat(1,3) if isotropic] + scatter(ds -> dygs);
 *     rotations = normalize(_rotation) + scatter(dr -> dygs);   opacities = sigmoid(_opacity);   shs = cat(_features_dc, _features_rest)
 * gsr_forward_raw / gsr_backward_raw read the raw tensors directly and apply these maps (and their chain rules) inside the
 * preprocess / geometry-backward kernels. Everything else is gsr_forward / gsr_backward_fused. */
typedef struct gsr_raw_inputs {
    const float* xyz;            /* [P,3]   GaussianModel._xyz */
    const float* log_scales;     /* [P,scale_dim]  _scaling (log space) */
    int scale_dim;               /* 3, or 1 for an isotropic model (gaussian_renderer/__init__.py:122-125) */
    const float* raw_rotations;  /* [P,4]   _rotation; normalised like torch.nn.functional.normalize (eps 1e-12) */
    const float* logit_opacity;  /* [P]     _opacity */
    const float* features_dc;    /* [P,1,3] _features_dc */
    const float* features_rest;  /* [P,M-1,3] _features_rest; may be NULL when M == 1 */
    const int* dyn_slot;         /* [P] or NULL: >= 0 selects the row of dx / ds / dr added to this Gaussian (position of the
                                    Gaussian inside pc.dygs), < 0 = static */
    const float* dx;             /* [K,3] or NULL */
    const float* ds;             /* [K,3] or NULL */
    const float* dr;             /* [K,4] or NULL */
    const int* gather;           /* [P] or NULL: rasterized Gaussian i is row gather[i] of the tensors above -- the boolean `mask`
                                    selection of render() (gaussian_renderer/__init__.py:179-191), P = number of selected rows */
    /* Flow mode (flow_proj1 != NULL): render_flow() of gaussian_renderer/__init__.py:229-361. The rasterized colour of a Gaussian is
     *   (ndc(xyz* + dx2[slot]; flow_proj2) - ndc(xyz* + dx[slot]; flow_proj1)).xy, 1 if slot >= 0 else 0)          (:262-284)
     * with ndc(p; M) = (p_h M).xy / ((p_h M).w + 1e-7) and xyz* the DETACHED position: the colour's gradient reaches dx and dx2 only,
     * the geometric gradient of the mean reaches xyz and dx as usual (:261,305). Opacity and the bases of scale / rotation are
     * constants in this mode (:307,326-334): only the ds / dr gradients are meaningful; features_* are not read (pass NULL, M = 1, D = 0). */
    const float* flow_dx2;       /* [K,3] second displacement (d_xyz2), or NULL = zero */
    const float* flow_proj1;     /* [4,4] full_proj_transform of camera 1 (row-vector convention, like projmatrix) */
    const float* flow_proj2;     /* [4,4] of camera 2 */
    /* Deformation-network deltas (render(dynamic=True), gaussian_renderer/__init__.py:149-157: the 4DGaussians deform_network returns
     * means3D + dx, _scaling + ds, _rotation + dr and render() applies the activations to THOSE): delta_mode = 1 adds dx / ds / dr in
     * front of the activations -- scales = exp(_scaling + ds), rotations = normalize(_rotation + dr) -- instead of behind them
     * (delta_mode = 0, the control-node deltas of :159-174). With delta_mode = 1 dyn_slot may be NULL: Gaussian i then takes row i.
     * The gradients of dx / ds / dr follow the same convention. Not combined with the flow mode.
     * delta_stride: floats between consecutive rows of dx, of ds and of dr, and of their gradients (0 = compact: 3, 3, 4) -- with
     * delta_stride = 10 the three are the column ranges [0,3), [3,6), [6,10) of the network's [K, 10] output (include/deformation_field.h). */
    int delta_mode;
    int delta_stride;
} gsr_raw_inputs;

typedef struct gsr_raw_grads {   /* all fully written (with `gather`: only the selected rows -- zero-fill them first); dx / ds / dr may be
                                    NULL (required when the corresponding input was given and its gradient is wanted) */
    float* xyz;                  /* [P,3] */
    float* log_scales;           /* [P,scale_dim] */
    float* raw_rotations;        /* [P,4] */
    float* logit_opacity;        /* [P] */
    float* features_dc;          /* [P,1,3] */
    float* features_rest;        /* [P,M-1,3] (NULL allowed when M == 1) */
    float* dx; float* ds; float* dr;   /* [K,3], [K,3], [K,4]: rows of slots that no Gaussian refers to are left untouched */
    float* dx2;                  /* [K,3] flow mode: gradient of flow_dx2 (NULL allowed) */
} gsr_raw_grads;

/* gsr_forward with the inputs described by `in` (no colors_precomp / cov3D_precomp / prefiltered in this variant). */
int gsr_forward_raw(gsr_alloc_fn geometry_alloc, void* geometry_user,
                    gsr_alloc_fn binning_alloc, void* binning_user,
                    gsr_alloc_fn image_alloc, void* image_user,
                    int P, int D, int M, const float* background, int width, int height,
                    const gsr_raw_inputs* in, float scale_modifier,
                    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                    float* out_color, float* out_depth, float* out_opacity, int* radii, int* n_touched,
                    int debug, void* stream);

/* Backward of gsr_forward_raw: dL_dmean2D[P,3] as in gsr_backward (with `gather`: one row per ROW of the raw tensors, like `out`), the parameter gradients of `out`, dL_dtau_sum[6] (may be NULL). */
int gsr_backward_raw(int P, int D, int M, int R,
                     const float* background, int width, int height,
                     const gsr_raw_inputs* in, float scale_modifier,
                     const float* viewmatrix, const float* projmatrix, const float* projmatrix_raw, const float* campos,
                     float tan_fovx, float tan_fovy, const int* radii,
                     char* geom_buffer, char* binning_buffer, char* image_buffer,
                     const float* dL_dpix, const float* dL_dpix_depth,
                     float* dL_dmean2D, const gsr_raw_grads* out, float* dL_dtau_sum,
                     int debug, void* stream);

/* ---- multi-view entry point: the views of ONE mapping iteration in one launch per pipeline stage ---------------------------------
 * The mapping back-end renders the same Gaussians from every window keyframe plus two random ones and back-propagates all of them
 * before one optimizer step (utils/slam_backend.py:357,526,657,768-771). gsr_forward_views / gsr_backward_views take the raw model
 * parameters once (gsr_raw_inputs; `gather` is not available here, the flow mode is chosen per call through the views) and V <= GSR_MAX_VIEWS view descriptors; every
 * stage of the pipeline is launched once with the view as a second grid dimension. Results per view are those of gsr_forward_raw /
 * gsr_backward_raw; the parameter gradients of `out` are the sum over the views, added in view order with one rounding per view --
 * with GSR_BACKWARD_ACCUMULATE exactly what V consecutive gsr_backward_raw calls in accumulate mode leave in the buffers.
 * Binning capacities are speculated per view slot (the v-th view of consecutive calls is assumed to look alike); the first call of a
 * slot, debug mode and frames that outgrow their capacity go through the single-view path inside the call. */
#define GSR_MAX_VIEWS 12
typedef struct gsr_view {
    const float* viewmatrix; const float* projmatrix; const float* projmatrix_raw; const float* cam_pos;   /* [16], [16], [16], [3] */
    const float* dx; const float* ds; const float* dr;   /* this view's deltas of the dynamic subset ([K,3], [K,3], [K,4]) or NULL; the
                                                            dx / ds / dr of the shared descriptor are ignored, its dyn_slot applies */
    float* out_color; float* out_depth; float* out_opacity; int* radii; int* n_touched;      /* forward outputs, as in gsr_forward */
    void* geometry_user; void* binning_user; void* image_user;                              /* user pointers of this view's allocations */
    char* geom_buffer; char* binning_buffer; char* image_buffer; int num_rendered;          /* filled by gsr_forward_views; pass back unchanged */
    const float* dL_dcolor; const float* dL_ddepth;                                         /* backward: cotangents [3,H,W], [1,H,W] */
    float* dL_dmean2D; float* ddx; float* dds; float* ddr; float* dL_dtau_sum;              /* backward: per-view gradients ([P,3]; deltas; [6]); ddx / dds / ddr / dL_dtau_sum may be NULL */
    /* flow views (render_flow, see gsr_raw_inputs.flow_*): set for EVERY view of a call or for none. The colour of such a view is the NDC
     * flow of the dynamic subset between (x + dx; flow_proj1) and (x + flow_dx2; flow_proj2) plus the mask channel; use D = 0, M = 1, a zero
     * background; features_dc of the shared descriptor may be NULL. In gsr_backward_views only out->xyz is required then (the other
     * parameters are constants of render_flow); ddx2 [K,3] receives the gradient of flow_dx2 (may be NULL). */
    const float* flow_dx2; const float* flow_proj1; const float* flow_proj2; float* ddx2;
    /* flow views whose caller only reads part of the image (the flow loss of utils/slam_backend.py:479-509 is masked to the keyframe's moving
     * pixels): DEVICE pointer to four ints, the tile rectangle [x0, y0, x1, y1) (16-pixel tiles, half open) this view is read in, or NULL = the
     * whole image. A Gaussian's tile rectangle is clipped to it (no instance in a tile outside; none at all -- radius 0, no gradient -- when
     * nothing is left): every pixel inside the rectangle is unchanged, bit for bit, and every gradient of a loss that only reads such pixels up
     * to the order in which a Gaussian's (now fewer) instance slots are added (~1e-9 relative); pixels outside are undefined (whatever the
     * remaining Gaussians leave). The pointer is read by the kernels: a captured call keeps reading the same address on replay. */
    const int* flow_clip;
} gsr_view;
int gsr_forward_views(int V, gsr_view* views, gsr_alloc_fn geometry_alloc, gsr_alloc_fn binning_alloc, gsr_alloc_fn image_alloc,
                      int P, int D, int M, const float* background, int width, int height, const gsr_raw_inputs* in, float scale_modifier,
                      float tan_fovx, float tan_fovy, int debug, void* stream);
/* scratch: device memory of gsr_views_scratch_size() bytes (one row of parameter gradients per view); not needed with GSR_BACKWARD_POSE_ONLY */
size_t gsr_views_scratch_size(int V, int P, int M, int scale_dim);
int gsr_backward_views(int V, gsr_view* views, int P, int D, int M, const float* background, int width, int height,
                       const gsr_raw_inputs* in, float scale_modifier, float tan_fovx, float tan_fovy,
                       const gsr_raw_grads* out, char* scratch, int debug, void* stream);

/* Rasterizer::markVisible (rasterizer.h:20-22 / rasterizer_impl.cu:54-66,141-153): present[i] = (z_view > 0.2). */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* stream);

/* Scratch sizes, the counterpart of required<GeometryState|ImageState|BinningState>() (rasterizer_impl.h:68-73).
 * gsr_forward passes exactly these sizes to the callbacks; exposed so a caller can pre-size arenas. */
size_t gsr_geometry_buffer_size(int P);
size_t gsr_image_buffer_size(int width, int height, int P);   /* P: the per-(Gaussian block, tile) binning scratch lives here */
size_t gsr_binning_buffer_size(int R_alloc);

/* Reads back intermediate state for stage-by-stage parity checks (tests only; synchronises `stream`).
 * Any destination may be NULL. depths[P], means2D[P,2], conic_opacity[P,4], rgb[P,3], cov3D[P,6], clamped[P,3] (0/1),
 * tiles_touched[P], point_offsets[P] (inclusive scan), final_T[H*W], n_contrib[H*W], ranges[T,2],
 * point_list (Gaussian id per sorted instance, packed tile after tile: exactly R entries). All HOST pointers. */
int gsr_debug_read_state(int P, int R, int width, int height,
                         const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                         float* depths, float* means2D, float* conic_opacity, float* rgb, float* cov3D, unsigned char* clamped,
                         uint32_t* tiles_touched, uint32_t* point_offsets, float* final_T, uint32_t* n_contrib,
                         uint32_t* ranges, uint32_t* point_list, void* stream);

/* Test hook for the transposed wave reduction used by the backward tile kernel: in = device float[64][10] (lane-major),
 * out = device float[64]; lane l receives the 64-lane total of value k(l): bit 1 of l set -> (bit 5 ? 5 : 0), else
 * (bit 4 ? 6 : 1) + bit 0 + 2 * bit 5. */
int gsr_debug_wave_reduce10(const float* in, float* out, void* stream);
/* Test hook, host only: which render_bwd block takes a work item -- the full piece of rank `rank` in tile order (partial = 0) or the partial
 * piece of rank `rank` in descending length (partial = 1) -- in a frame of n_items pieces, n_partial of them partial. The map is a bijection
 * onto [0, n_items); block b runs on XCD b % 8 in the order of b / 8 (csrc/gs_device.h: item_block_*). Negative: argument out of range. */
int gsr_debug_item_block(unsigned int n_items, unsigned int n_partial, unsigned int rank, int partial);

/* Per-host-thread options of the forward pass; returns the previous value (value < 0: query only) or a negative error code.
 *   "speculate" (default 1): scatter / sort / render are enqueued on a binning buffer sized from the previous frame before the host
 *                knows num_rendered; 0 = wait for it first, like the reference's blocking copy (rasterizer_impl.cu:283-284).
 *   "lazy"      (default 0): a speculative gsr_forward returns without ANY host wait; its return value is then an upper bound of
 *                num_rendered (the buffer capacity) -- valid as the R argument of gsr_backward, whose kernels read the true counts
 *                on the device. Needed to capture forward + backward in a hipGraph. A frame that outgrows the capacity leaves its
 *                outputs undefined and bumps the overflow counter of gsr_forward_status(): poll it and redo that work with lazy = 0.
 *   "mailbox"   (default 1): read the header through pinned host memory instead of a blocking copy.
 *   "cap_margin_permille" (default 125): head room of a speculative binning buffer over the previous frame's instance count, in
 *                1/1000. A caller that captures many iterations of an OPTIMISATION in a hipGraph (slam/mapping_graph.py: the buffer
 *                is laid out once, the Gaussians then move and grow for 20-200 replays) raises it for the capture.
 *   "cap_tile_margin_permille" (default: max(250, cap_margin_permille)): head room of the LONGEST TILE LIST over the previous frame's,
 *                which selects the sort kernels a speculative forward pass enqueues. A tile list can double where the instance count
 *                moves by a few percent (a dynamic object's Gaussians piling up while the node network trains): captures raise it
 *                separately. The value 1000000 stands for the default, when set and when returned.
 *   "cap_floor" (default 0): smallest capacity, in instances, a speculative binning buffer is laid out for whatever the previous frame needed.
 *                A view slot that a captured iteration fills with a DIFFERENT keyframe on every replay (the two random keyframes of a mapping
 *                iteration and their flow renders, utils/slam_backend.py:1031-1037) has no meaningful "previous frame": the estimate is one
 *                candidate's count, another candidate may need three times as much (measured: 7 042 instances at capture, > 25 000 at a later
 *                replay). Such captures set a floor (a few MB per view) instead of trusting a ratio.
 *   "view_slot_group" (default 0; 0..3): which of an iteration's gsr_forward_views calls the next calls are. The capacity estimate and
 *                the mailbox of a view are kept per (group, flow / plain, position in the call): a caller that needs several calls per
 *                iteration (more than GSR_MAX_VIEWS flow renders) numbers them, or the v-th view of two calls would share -- and spoil -- one
 *                estimate.
 *   "order_items" (default 1): render_bwd's work items (128-entry pieces of the tile lists) are laid out so that every XCD runs the full
 *                pieces first, in tile order, and ends on the partial last pieces of the lists, longest first -- the launch then ends on
 *                short blocks everywhere at once instead of a ~20 us tail. One extra block of the scatter launch ranks the pieces. The
 *                results are bit-identical either way (every piece writes its own instances' slots); 0 = tile order. Read by the FORWARD
 *                pass (which writes the work-item table).
 *   "sh_rows" (default 1): the SH coefficients of a wave's 64 Gaussians move as whole rows through LDS (one DMA instruction / one store per
 *                row) instead of one strided access per coefficient and lane; same arithmetic, bit-identical results. 0 = per lane.
 *   "hex_ordered" (default 1; PROCESS-wide, value < 0 only reads): the HexPlane field's sorted backward passes (deformation_field.h) sum the plane
 *                gradients as fixed-point integers -- bitwise reproducible; 0 = float atomics. Environment: GSR_HEX_ORDERED.
 *   "cap_test_shrink_permille" (default 0 = off): TEST facility -- lay speculative buffers out for this fraction of the previous
 *                frame's count, so that overflows (and the callers' recovery paths) can be provoked deliberately.
 * Environment: GSR_SPECULATE, GSR_LAZY, GSR_MAILBOX, GSR_ORDER_ITEMS, GSR_SH_ROWS set the initial values. */
int gsr_set_option(const char* name, int value);
/* overflow_count: number of forward passes of this thread whose speculative capacity was too small (sticky);
 * last_num_rendered: num_rendered of the most recent forward pass the GPU has finished binning. Never blocks. */
int gsr_forward_status(unsigned int* overflow_count, unsigned int* last_num_rendered);
/* The same sticky counter summed over the single-view slot AND every view slot of gsr_forward_views / its flow batches (each slot has
 * its own mailbox) on the current device. Never blocks. */
int gsr_forward_status_views(unsigned int* overflow_count_total);

/* Thread-local text of the last error. */
const char* gsr_last_error(void);

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * gsr_profile_enable(mask) times the kernels whose bit is set in `mask` (bit i = i-th entry of gsr_profile_read;
 * 0 disables, -1 enables all) and returns the number of kernels; gsr_profile_read copies up to `cap` entries
 * (name pointers are static strings; ms = summed duration; calls = launches) and returns the count;
 * it synchronises the recorded events. gsr_profile_reset() clears the accumulators. */
int gsr_profile_enable(int kernel_mask);
int gsr_profile_read(const char** names, float* total_ms, int* calls, int cap);
void gsr_profile_reset(void);

const char* gsr_version(void);

#ifdef __cplusplus
}
#endif
#endif

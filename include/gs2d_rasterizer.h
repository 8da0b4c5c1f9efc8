/*
 * gs2d_rasterizer.h -- C ABI of the MI355X-native 2D-Gaussian-surfel rasterizer.
 *
 * Drop-in boundary for the hot path of vasabi-root/gaus-slam.  Every entry
 * point replaces one interface of the reference (paths relative to
 * /root/reference, RAST = submodules/gaus_2dgs_rasterization):
 *
 *   gs2d_forward       <- CudaRasterizer::Rasterizer::forward
 *                         RAST/cuda_rasterizer/rasterizer.h:42-71, rasterizer_impl.cu:201-350
 *   gs2d_backward      <- CudaRasterizer::Rasterizer::backward
 *                         RAST/cuda_rasterizer/rasterizer.h:73-107, rasterizer_impl.cu:354-460
 *   gs2d_mark_visible  <- CudaRasterizer::Rasterizer::markVisible
 *                         RAST/cuda_rasterizer/rasterizer.h:35-40, rasterizer_impl.cu:141-153
 *   sknn_dist2         <- simple_knn._C.distCUDA2 (call sites scene/Gaussians.py:77,218;
 *                         third-party module, source absent from the mounted reference)
 *
 * Conventions (identical to the reference unless stated):
 *   - all pointers are DEVICE pointers to float32 / int32 data, plain C layouts;
 *     NULL selects the alternative path exactly as the reference's nullptr does
 *     (colors_precomp vs shs, scales+rotations vs transMat_precomp);
 *   - viewmatrix / projmatrix are 16 floats, column-major (the transposed
 *     matrices render/render_2dgs.py:10-24 builds);
 *   - scratch memory is obtained through caller-supplied allocator callbacks
 *     (the C form of the reference's std::function<char*(size_t)>); the three
 *     chunks are OPAQUE and must stay alive, unwritten and at the same address
 *     until the matching gs2d_backward: the library keeps a host-side record per
 *     forward (keyed by the geometry chunk's address) that says which mode the
 *     binning chunk was sized for and that the gradient accumulator inside the
 *     geometry chunk is still zero.  A backward on a chunk without a record (a
 *     copy at another address, more than 64 forwards in flight) clears the
 *     accumulator itself and cannot run in deterministic mode; a backward whose
 *     R / binning pointer / deterministic flag differ from its forward's returns
 *     an error instead of touching memory.  The internal layout is private
 *     (query it with the gs2d_*_layout helpers, used by the parity tests only);
 *   - `stream` is a hipStream_t (NULL = the null stream).  The reference used
 *     the legacy default stream; callers here pass torch's current HIP stream;
 *   - return value < 0 signals an error; gs2d_last_error() describes it.
 *     (The reference throws std::runtime_error / AT_ERROR.)
 *
 * Deliberately dropped from the reference signatures: forward's `out_mask` ([1,H,W], allocated by
 * RasterizeGaussiansCUDA -- rasterize_points.cu:89 -- and passed to Rasterizer::forward, rasterizer.h:42-71, but never
 * written or read by any kernel), the unused `focal_x/focal_y` blend arguments and `tan_fov*` of the forward preprocess
 * (forward.cu:264-268) -- tan_fovx/tan_fovy are still accepted here because the BACKWARD rebuilds W and H from them
 * (backward.cu:641-642).  The per-stage timing helpers (gs2d_stage_timing_*) are a bench facility: process-global and not
 * thread-safe.
 *
 * No torch types appear in this header.  The reference-side binding a
 * maintainer would write is shown in INTEGRATION.md.
 */
#ifndef GS2D_RASTERIZER_H
#define GS2D_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scratch allocator: must return a device pointer to >= `bytes` bytes, 256-byte aligned.  A callback may be invoked more
 * than once per call (the binning chunk is requested early from an estimate and again if that was too small); only the
 * pointer returned LAST is handed on to the backward, and the chunk may be larger than the library strictly needs.  A kernel
 * enqueued on `stream` may still write into the EARLIER chunk when the later request arrives (the forward fills the estimated
 * chunk before it knows num_rendered): an allocator that recycles the earlier chunk must do so in the order of that stream,
 * as a stream-ordered allocator (torch's caching allocator, hipMallocAsync / hipFreeAsync on `stream`) does by itself. */
typedef void* (*gs2d_alloc_fn)(void* user, size_t bytes);

/* Returns num_rendered (>= 0, number of (tile, Gaussian) instances) or < 0 on error.
 * Performs one stream synchronisation (as the reference does, rasterizer_impl.cu:287). */
int gs2d_forward(
    gs2d_alloc_fn geometry_alloc, void* geometry_user,
    gs2d_alloc_fn binning_alloc, void* binning_user,
    gs2d_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background,          /* [3] */
    int width, int height,
    const float* means3D,             /* [P,3] */
    const float* shs,                 /* [P,M,3] or NULL */
    const float* colors_precomp,      /* [P,3]  or NULL */
    const float* opacities,           /* [P] */
    const float* scales,              /* [P,2]  or NULL */
    float scale_modifier,
    const float* rotations,           /* [P,4] (w,x,y,z) or NULL */
    const float* transMat_precomp,    /* [P,9]  or NULL */
    const float* viewmatrix,          /* [16] column-major */
    const float* projmatrix,          /* [16] column-major */
    const float* cam_pos,             /* [3] */
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float* out_color,                 /* [3,H,W] */
    float* out_others,                /* [7,H,W]: depth, alpha, normal xyz, median depth, dist */
    int* radii,                       /* [P] */
    int use_sa, int debug, void* stream);

/* Every element of every dL_* output is written (zeros for Gaussians culled by the forward), so the caller may
 * pass uninitialised memory; the reference relies on torch::zeros instead (rasterize_points.cu:192-200).
 * scale_modifier: the value the forward was called with, as the reference's autograd function passes it
 * (RAST/gaus_2dgs_rasterization/__init__.py:67 and :116, the same raster_settings).  The rebuilt transform ignores it (backward.cu:504); the library uses it
 * to know whether the forward's Tw.z can be recomputed (== 1) or has to be read from the geometry chunk (!= 1). */
int gs2d_backward(
    int P, int D, int M, int R,
    const float* background,
    int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations,
    const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    char* geom_buffer, char* binning_buffer, char* img_buffer,
    const float* dL_dpix,             /* [3,H,W] */
    const float* dL_depths,           /* [7,H,W] */
    float* dL_dmean2D,                /* [P,3] */
    float* dL_dnormal,                /* [P,3] or NULL: internal in the reference (never returned to Python) */
    float* dL_dopacity,               /* [P]   */
    float* dL_dcolor,                 /* [P,3] */
    float* dL_dmean3D,                /* [P,3] */
    float* dL_dtransMat,              /* [P,9] or NULL when the caller has no use for it (no cov3D_precomp input) */
    float* dL_dsh,                    /* [P,M,3] */
    float* dL_dscale,                 /* [P,2] */
    float* dL_drot,                   /* [P,4] */
    int use_sa, int debug, void* stream);

/*
 * Tracking-regime variants (no counterpart in the reference's native API -- the reference does
 * this part in PyTorch, render/__init__.py:31-40): the rigid camera transform is applied INSIDE the preprocess
 * kernels,  means3D_cam = R x + t,  rotations = standardize(q_cam (x) q)  (pytorch3d quaternion_multiply, real part
 * first), and the backward additionally returns the pose gradient
 *     dL_dpose[12] (row-major [dL/dR | dL/dt]),  dL/dR = sum_i g_i (x) x_i,  dL/dt = sum_i g_i,
 * with dL_dmean3D = R^T g_i and dL_drot mapped back to the untransformed quaternions.
 * pose_Rt: 12 floats row-major [R | t] (device); pose_quat: q_cam (w,x,y,z) (device).  Both NULL = plain call.  pose_quat alone
 * may be NULL: the kernels then derive it from pose_Rt's rotation block themselves (gs2d_pose_quat's code, same bits; no
 * extra launch in front of every tracking iteration); a caller with a quaternion of its own passes it.
 * Typical use: identity viewmatrix / projection of the intrinsics only, as the reference's tracking renderer does.
 * Pose-only backward: with a pose (and no SH), dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dscale and dL_drot may
 * ALL be NULL -- tracking detaches every Gaussian parameter (render/__init__.py:31-36), only dL_dpose is produced then.
 */
/* pose_quat for the calls below, computed on the device from pose_Rt's rotation block (pytorch3d's
 * matrix_to_quaternion, real part first, w >= 0): quat_out = 4 floats (device).  No host synchronisation. */
int gs2d_pose_quat(const float* pose_Rt, float* quat_out, void* stream);

int gs2d_forward_posed(
    gs2d_alloc_fn geometry_alloc, void* geometry_user, gs2d_alloc_fn binning_alloc, void* binning_user,
    gs2d_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
    float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
    float* out_others, int* radii, int use_sa, int debug, const float* pose_Rt, const float* pose_quat, void* stream);

int gs2d_backward_posed(
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D, const float* shs,
    const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
    const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
    float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix,
    const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot, int use_sa, int debug, const float* pose_Rt,
    const float* pose_quat, float* dL_dpose /* [12] */, void* stream);

/*
 * The backward in two stages, the second one on a range of Gaussians (no counterpart in the reference, whose backward is one
 * call -- rasterizer_impl.cu:354-460): stage GS2D_BWD_BLEND clears the per-Gaussian gradient records and runs the blend
 * backward over the whole image; stage GS2D_BWD_PREPROCESS turns the records of Gaussians [g_begin, g_end) into the dL_*
 * outputs (same full-size output pointers; only rows of that range are written).  A caller that shards keyframes over GPUs
 * (gaus_slam_amd/ba_shard.py) runs BLEND once and PREPROCESS chunk by chunk, starting the gradient all-reduce of a chunk
 * while the next chunk is still being computed.  gs2d_backward_posed == stages 3 on [0, P).  dL_dpose accumulates over the
 * PREPROCESS calls and is cleared by the BLEND call.
 */
#define GS2D_BWD_BLEND 1
#define GS2D_BWD_PREPROCESS 2
/* Together with both stages: dL_dpose points at SIXTEEN floats, a row-major 4x4 whose first three rows receive [dL/dR | dL/dt]
 * and whose fourth row is cleared by the call as well -- the gradient of a [4,4] world-to-camera matrix
 * (render/__init__.py:31-40 optimises such a matrix) without a separate fill of the buffer. */
#define GS2D_BWD_POSE_4X4 4
int gs2d_backward_staged(
    int stages, int g_begin, int g_end,
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D, const float* shs,
    const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
    const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
    float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix,
    const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot, int use_sa, int debug, const float* pose_Rt,
    const float* pose_quat, float* dL_dpose /* [12] */, void* stream);

/*
 * Batched keyframes (no counterpart in the reference, whose backend renders one keyframe per step, slam/Backend.py:101-128):
 * K frames of the same size over the SAME Gaussians -- what a bundle-adjustment rank holds when keyframes outnumber GPUs
 * (gaus_slam_amd/ba_shard.py).  Per frame the calls do exactly what gs2d_forward / gs2d_backward do, with the same
 * per-frame scratch chunks and bit-identical per-frame outputs; the two blend passes run as ONE grid over the tiles of all
 * K frames, so the next frame's tiles fill the SIMDs the previous frame's last waves leave idle, and the host waits once
 * for all K num_rendered values.  Mapping / BA regime only: no fused pose, no deterministic variant (use the per-frame
 * calls for those).  1 <= K <= GS2D_MAX_FRAMES.
 */
#define GS2D_MAX_FRAMES 8
typedef struct gs2d_frame_io {       /* per-frame arguments of gs2d_forward (same meaning) */
    gs2d_alloc_fn geometry_alloc; void* geometry_user;
    gs2d_alloc_fn binning_alloc; void* binning_user;
    gs2d_alloc_fn image_alloc; void* image_user;
    const float* viewmatrix; const float* projmatrix; const float* cam_pos;
    float* out_color;                /* [3,H,W] */
    float* out_others;               /* [7,H,W] */
    int* radii;                      /* [P] */
} gs2d_frame_io;
/* num_rendered[k] receives frame k's instance count.  Returns 0 or < 0 on error. */
int gs2d_forward_batch(
    int K, const gs2d_frame_io* frames, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
    float scale_modifier, const float* rotations, const float* transMat_precomp, int use_sa, int debug,
    int* num_rendered /* [K], host */, void* stream);

typedef struct gs2d_frame_grad {     /* per-frame arguments of gs2d_backward (same meaning) */
    const float* viewmatrix; const float* projmatrix; const float* campos;
    float tan_fovx, tan_fovy;
    const int* radii;
    char* geom_buffer; char* binning_buffer; char* img_buffer;
    int num_rendered;
    const float* dL_dpix;            /* [3,H,W] */
    const float* dL_depths;          /* [7,H,W] */
    float* dL_dmean2D; float* dL_dnormal; float* dL_dopacity; float* dL_dcolor; float* dL_dmean3D;
    float* dL_dtransMat; float* dL_dsh; float* dL_dscale; float* dL_drot;
} gs2d_frame_grad;
/* accumulate != 0: after the per-frame gradients have been written, frame 0's PARAMETER gradients (dL_dmean3D, dL_dcolor,
 * dL_dopacity, dL_dscale, dL_drot, dL_dsh, dL_dnormal, dL_dtransMat) receive the sum over all K frames (added in frame order:
 * the result K separate backwards followed by tensor additions give) -- what a BA rank needs; frames 1 .. K-1 still hold
 * their own gradients.  dL_dmean2D is NEVER summed: the screen-space gradient is a per-view quantity (the reference
 * accumulates its norm view by view for densification, scene/Gaussians.py:58-62), every frame keeps its own. */
int gs2d_backward_batch(
    int K, const gs2d_frame_grad* frames, int accumulate, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
    const float* rotations, const float* transMat_precomp, int use_sa, int debug, void* stream);

/*
 * Deterministic backward (opt-in, process-wide; off by default).  The reference accumulates per-Gaussian gradients with
 * float atomics (backward.cu:343,396,441-460), so its gradients differ from run to run in the last bits, and so do this
 * library's by default.  With the switch on, the backward uses no atomics: every (instance, quadrant) pair is summed by
 * one wave into a record of its own and a Gaussian's records are added in a fixed order -- two runs on the same inputs give
 * bit-identical gradients (tests/test_gpu_round2.py).  Slower (about 2x for the backward) and 324 B more scratch per tile
 * instance.  The forward sizes the binning chunk for the mode it runs in and records that mode; a backward called with the
 * switch in the other position fails with an error (nothing is launched) -- restore the switch or rerun the forward.
 * The POSE gradient (dL_dpose of the *_posed / *_staged calls) is deterministic in this mode as well: one partial per workgroup
 * of the per-Gaussian stage, summed in a fixed order by a reduction kernel (tests/test_gpu_round3.py); the partials are kept
 * in the geometry chunk's `depths` array, which therefore no longer holds the view depths after a deterministic posed backward.
 */
void gs2d_set_deterministic(int on);
int gs2d_get_deterministic(void);

/*
 * Tile binning mode (process-wide; default 0).  The reference makes one (tile, Gaussian) instance for every tile of the
 * square around a Gaussian's 3-sigma radius (rasterizer_impl.cu:70-111, auxiliary.h:66-76).  By default this library only
 * makes the instances inside the splat's footprint bound -- the region where alpha can reach 1/255 at all, which depends on
 * the opacity and follows the projected ellipse; every instance left out is one whose pixels the reference would all
 * `continue` past (forward.cu:385-387), so colours, depth maps and gradients are unchanged (the forward outputs are
 * bit-identical between the two modes, tests/test_gpu_footprint.py) while the sort and both blend passes handle about a
 * fifth fewer instances.  Consequences of the default: num_rendered, tiles_touched and the per-tile lists are the
 * reference's minus those instances (the lists are ordered subsequences of the reference's).  radii are the reference's.
 * gs2d_set_reference_binning(1) switches to the reference's rectangles: then num_rendered and the sorted lists are
 * bit-identical to the reference's (what the binning parity tests compare against the oracle).
 */
void gs2d_set_reference_binning(int on);
/* Launch-ahead forward (default on): a single-frame forward enqueues ALL its kernels before the host looks at num_rendered --
 * the stages behind the instance duplication read the count on the device, their grids are sized for the capacity of the
 * binning chunk (requested from the previous call's count for the same problem shape, + 12.5 %), and the one host read of the
 * reference's forward (rasterizer_impl.cu:287) happens after the blend kernel is enqueued, when the GPU's queue is full.
 * A count beyond the capacity (first call of a shape, a scene that grew by more than 12.5 % between calls) makes those
 * kernels do nothing and the host run the stages again in a chunk of the exact size, in stream order: results never differ.
 * 0 restores the round-3 order (duplicate, host wait, the rest); debug mode, the deterministic mode, batches and images of
 * more than 4096 tiles always use that order.  Process-wide. */
void gs2d_set_launch_ahead(int on);
int gs2d_get_launch_ahead(void);
int gs2d_get_reference_binning(void);

/* present: [P] bytes (0/1). */
int gs2d_mark_visible(int P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, uint8_t* present, void* stream);

/* Mean squared distance to the 3 nearest other points; points [N,3], out [N].
 * `ws_alloc` provides temporary device memory (may be freed after the stream drains). */
int sknn_dist2(int N, const float* points, float* out,
               gs2d_alloc_fn ws_alloc, void* ws_user, void* stream);

/*
 * Fused post-op + loss of the SLAM iterations (the reference does this in PyTorch:
 * render/__init__.py:46-49 weight-normalised depth + outlier zeroing, slam/Loss.py:22-58 nan_to_num, masks, masked L1).
 * mode 0 = tracking (masked sums), 1 = mapping (masked means + dist term).  color [3,H,W], allmap [7,H,W] are the raw
 * rasterizer outputs, gt_color_hwc [H,W,3], gt_depth [H,W].  Writes loss_out[0] = loss (loss_out[1..5] = colour sum,
 * depth sum, dist sum, #colour-mask, #depth-mask) and the gradients of the loss w.r.t. color / allmap.
 * Two-phase use (what the autograd node does): call with dL_dcolor = dL_dallmap = NULL for the loss alone, later with
 * loss_out = NULL (same inputs, same workspace) for the gradients, scaled by the device scalar *upstream.
 * Default-configuration losses only (no normal loss, no outlier rejection, no exposure).
 */
#define GS2D_LOSS_WS_DOUBLES 2560
int gs2d_slam_loss(int mode, int width, int height, const float* color, const float* allmap, const float* gt_color_hwc,
                   const float* gt_depth, float w_color, float w_depth, float w_dist, float silmask_th, float edge_thres,
                   int use_edge_growth, int use_weight_norm, float eps, float depth_near, float depth_far,
                   double* workspace /* >= GS2D_LOSS_WS_DOUBLES doubles, need not be initialised */, float* loss_out /* [8] */,
                   float* dL_dcolor, float* dL_dallmap, const float* upstream /* device scalar dL/dloss or NULL (= 1) */,
                   void* stream);

/*
 * Fused dense Adam over the flat Gaussian SoA (the reference: torch.optim.Adam(l, lr=0.0, eps=1e-15)
 * over five tensors, scene/Gaussians.py:121-137).  param / grad / exp_avg / exp_avg_sq are flat fp32 device buffers of n
 * elements (16-byte aligned); group g covers elements [group_end[g-1], group_end[g]) and uses learning rate group_lr[g]
 * (group_end / group_lr are HOST arrays, n_groups <= GS2D_ADAM_MAX_GROUPS, group_end[n_groups-1] == n).  `step` is the
 * 1-based step count used for the bias corrections.  No weight decay, no amsgrad.
 */
#define GS2D_ADAM_MAX_GROUPS 8
int gs2d_adam_step(int n_groups, const unsigned long long* group_end, const float* group_lr, float beta1, float beta2,
                   float eps, int step, unsigned long long n, float* param, const float* grad, float* exp_avg,
                   float* exp_avg_sq, void* stream);

/* Sizes of the three scratch chunks (what the allocator callbacks will be asked for). */
size_t gs2d_geometry_bytes(int P);
size_t gs2d_image_bytes(int width, int height);
size_t gs2d_binning_bytes(int R);

/*
 * Private-layout introspection for the parity tests: byte offsets of the
 * sub-arrays inside each chunk.  Geometry: [0] depths f32[P], [1] tiles_touched
 * u32[P], [2] point_offsets u32[P], [3] splat records f32[P][20], [4] clamped
 * u8[3P].  Binning: [0] point_list u32[R], [1] sorted keys u64[R].
 * Image: [0] ranges u32[tiles][2], [1] pixel state f32/u32[7][tiles*256]
 * (planes: T_final, M1, M2, median depth, depth std, last contributor, median
 * contributor; element index = tile*256 + quadrant*64 + group*4 + (y%2)*2 + (x%2) with
 * quadrant = (y/8)*2 + (x/8) and group = ((y%8)/2)*4 + (x%8)/2 for pixel (x,y) inside its 16x16 tile).
 */
void gs2d_geometry_layout(int P, size_t offsets[5]);
void gs2d_binning_layout(int R, size_t offsets[2]);
void gs2d_image_layout(int width, int height, size_t offsets[2]);

/* Optional per-stage device timing with hipEvents recorded on the launch stream (bench.py's roofline leg).
 * ms[9] = preprocess, scan, duplicate, sort, ranges, blend_fwd, blend_bwd, preprocess_bwd, cull; -1 = not recorded
 * (cull: always -1, the sub-block cull runs as the first phase of blend_fwd). */
void gs2d_stage_timing_enable(int on);
int gs2d_stage_timing_read(float ms[9]);
/* ms[2 i], ms[2 i + 1] = begin / end of stage i relative to the begin of the latest preprocess stage (-1: not recorded). */
int gs2d_stage_timing_read_abs(float ms[18]);

const char* gs2d_last_error(void);
const char* gs2d_build_info(void);

#ifdef __cplusplus
}
#endif
#endif

/*
 * srh.h -- C ABI of the MI355X-native ("hip") backend for DiffRend's render(scene) hot path.
 *
 * The reference project (fmannan/surf_renderer) has no FFI for this path: each backend is a Python
 * module exposing `render(scene) -> dict` (diffrend/numpy/renderer.py:204-272,
 * diffrend/torch/renderer.py:136-355).  This header is the boundary a `--use hip` backend binds
 * instead: one call per frame per GPU, plain pointers and sizes, no torch / numpy types.  The
 * Python mirror of the reference interface lives in surf_renderer_amd/renderer.py and calls these
 * entry points through ctypes (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *   - Every device buffer is allocated and freed by the caller.  The library keeps no device memory
 *     and no state between calls; all work is enqueued on the stream that is passed in and no entry
 *     point synchronises.  One documented exception, srh_render_views: its per-view frame descriptors
 *     go through pinned staging buffers and a ring of four batches in constant memory that the library
 *     owns PER DEVICE (created on first use of a device, kept until the process ends, one mutex per
 *     device; calls on different devices do not share anything).  A call waits on the host only if the
 *     slot it reuses is still in flight, four calls back; it must be made with the stream's device
 *     current, and it cannot be stream-captured (srh_render_fwd can).
 *   - Thread safety: every entry point may be called from any thread; srh_last_error() is per thread.
 *     Concurrent srh_render_views calls on one device serialise on that device's mutex.
 *   - Arrays use the reference's layouts (docs/scene_description.md, numpy/renderer.py:299-358):
 *     homogeneous 4-vectors, points w = 1, directions / normals w = 0, row-major, float32 on the
 *     device; index arrays are int32.
 *   - Primitive numbering is the reference's: segments in scene['objects'] dict order, running offset
 *     (numpy/renderer.py:172-201).  The lowest global index wins exact depth ties (np.argmin, :223).
 *   - Return value: 0 on success, a positive hipError_t, or a negative SRH_E_* code.  No C++ exception
 *     crosses the boundary; srh_last_error() returns a thread-local message for the last failure.
 */
#ifndef SRH_H
#define SRH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRH_ABI_VERSION 10
#define SRH_MAX_SEGMENTS 4
#define SRH_MAX_LIGHTS 64

/* primitive types: keys of scene['objects'] (numpy/renderer.py:133-137) */
enum { SRH_PRIM_DISK = 0, SRH_PRIM_PLANE = 1, SRH_PRIM_SPHERE = 2, SRH_PRIM_TRIANGLE = 3 };

/* error codes (negative; positive values are hipError_t) */
enum {
  SRH_OK = 0,
  SRH_E_NULL = -1,        /* a required pointer is NULL */
  SRH_E_RANGE = -2,       /* a count / row range / viewport is out of range */
  SRH_E_TYPE = -3,        /* unknown primitive type or mode */
  SRH_E_WORKSPACE = -4,   /* workspace too small or misaligned */
  SRH_E_CAMERA = -5       /* degenerate camera (eye == at, zero up, w conventions violated) */
};

/* render modes (SrhParams.mode) -- all modes produce bit-identical outputs */
enum {
  SRH_MODE_AUTO = 0,      /* the fastest exact path */
  SRH_MODE_EXACT = 1,     /* every (pixel, primitive) pair through the fp64 intersection (checker mode) */
  SRH_MODE_FAST = 2,      /* all pairs: fp32 conservative screen-space reject + fp64 confirmation of survivors */
  SRH_MODE_BINNED = 3     /* primitives binned to 16x16-pixel tiles by their screen bounding box, then as FAST */
};

/* scene['camera'] (numpy/renderer.py:145-169, numpy/ops.py:88-115).  Host memory, float64.
 * The caller applies the reference's float32 detour for list-typed at/up before filling this (see up_is_unit). */
typedef struct SrhCamera {
  double eye[4];          /* w must be 1 */
  double at[4];
  double up[4];           /* w must be 0 */
  double fovy;            /* radians */
  double focal_length;
  double near_clip;       /* valid hit: near <= t <= far on Euclidean ray distance (:219) */
  double far_clip;
  int32_t viewport[4];    /* x0, y0, x1, y1; W = x1 - x0, H = y1 - y0 */
  int32_t ortho;          /* 1: orthographic projection (torch/utils.py:461-468) -- SRH_SHADING_TORCH only (the numpy
                             backend has none); forward (all pairs in fp64), backward and srh_render_views; 0: perspective */
  int32_t up_is_unit;     /* 1: `up` is already the camera's y axis and is used as given.  The reference normalises a
                             list-typed `up` in float32 arithmetic (numpy/ops.py:99,109), which the host repeats with
                             the same numpy call before filling this; 0: y = up / |up| in float64 */
} SrhCamera;

/* one entry of scene['objects']: a batch of primitives of one type (device pointers) */
typedef struct SrhSegment {
  int32_t type;                 /* SRH_PRIM_* */
  int32_t count;
  const float* pos;             /* (count,4)   disk, plane, sphere */
  const float* normal;          /* (count,4)   disk, plane, triangle (never recomputed from vertices, :107) */
  const float* radius;          /* (count)     disk, sphere */
  const float* face;            /* (count,3,4) triangle */
  const int32_t* material_idx;  /* (count) */
} SrhSegment;

typedef struct SrhObjects {
  int32_t n_segments;           /* <= SRH_MAX_SEGMENTS, scene['objects'] dict order */
  SrhSegment seg[SRH_MAX_SEGMENTS];
} SrhObjects;

/* scene['lights'] + scene['colors'] (numpy/renderer.py:234-237) */
typedef struct SrhLights {
  int32_t n_lights;             /* <= SRH_MAX_LIGHTS */
  int32_t n_colors;
  const float* pos;             /* (n_lights,4) device */
  const int32_t* color_idx;     /* (n_lights)   device, rows of `colors` */
  const float* colors;          /* (n_colors,3) device */
  const float* attenuation;     /* (n_lights,3) device, (kc, kl, kq); SRH_SHADING_TORCH only, NULL = (1,0,0) */
  const float* ambient;         /* (3) device; SRH_SHADING_TORCH only, NULL = 0 */
} SrhLights;

/* scene['materials'] (numpy/renderer.py:245) */
typedef struct SrhMaterials {
  int32_t n_materials;
  const float* albedo;          /* (n_materials,3) device */
  const float* coeffs;          /* (n_materials,3) device, (diffuse, specular, shininess); SRH_SHADING_TORCH only,
                                   NULL = (1,0,0) */
} SrhMaterials;

/* Which of the reference's two differentiable-renderer semantics a frame follows (SrhParams.shading). */
enum {
  SRH_SHADING_NUMPY = 0,  /* diffrend/numpy/renderer.py: Lambert, clip after the light sum, +inf background, the
                             reference's non-orthonormal camera basis and 4-D normalisation */
  SRH_SHADING_TORCH = 1   /* diffrend/torch/renderer.py:82-125,136-355: attenuation, per-light relu, specular, ambient,
                             double_sided, use_quartic, orthonormal camera basis, far+1 background */
};

typedef struct SrhParams {
  int32_t row0, row1;           /* render image rows [row0,row1) of the camera's H rows; the output
                                   buffers hold only those rows (multi-GPU row slabs) */
  int32_t mode;                 /* SRH_MODE_* */
  int32_t tonemap_gamma;        /* 1: image <- image ** gamma  (scene has a 'tonemap' entry, :262) */
  double gamma;
  int32_t shading;              /* SRH_SHADING_* */
  int32_t double_sided;         /* torch shading: flip the normal towards the viewer (torch/renderer.py:107-112) */
  int32_t use_quartic;          /* torch shading: attenuation uses d^4 instead of d^2 (torch/renderer.py:92) */
  int32_t waves_per_tile;       /* binned mode: 0 = choose by tile count, 1 or 4 = force (tuning / tests; same result) */
  float* normal_out;            /* optional (rows,W,3) dense: unit normal of the hit, 0 where nothing is hit */
  float* pos_out;               /* optional (rows,W,3) dense: hit point, 0 where nothing is hit */
  int64_t image_row_stride;     /* elements between consecutive output rows; 0 = dense (3*W, W, W). */
  int64_t depth_row_stride;     /* Lets image and depth rows interleave in one (rows, 4*W) slab so that a */
  int64_t nearest_row_stride;   /* multi-GPU frame is collected by a single gather. */
  void* ev_start;               /* optional hipEvent_t pair recorded on `stream` immediately before and */
  void* ev_stop;                /* after the frame's dominant kernel (measurement hook); NULL = off */
  const void* visibility;       /* srh_render_bwd, SRH_SHADING_TORCH: the (rows,W) uint64 light-visibility bits that
                                   srh_shadow_shade wrote for this frame, or NULL (no shadows) */
  const int32_t* view_row0;     /* srh_render_views only: HOST array of n_views first rows; view v renders rows
                                   [view_row0[v], view_row0[v] + row1 - row0).  NULL = every view renders [row0, row1) */
  int32_t stages;               /* srh_render_fwd, binned mode: 0 = the whole frame; SRH_STAGE_BIN = only the per-frame
                                   records and tile bins (into the workspace); SRH_STAGE_RENDER = only the render kernel,
                                   from the bins a SRH_STAGE_BIN call with the same arguments left in the workspace.
                                   Lets a caller run the latency-bound binning of frame i+1 on one stream beside the
                                   render kernel of frame i on another (surf_renderer_amd/pipeline.py) */
  int32_t counters_clean;       /* binned frames: 1 = the caller KNOWS that this workspace's bin counters are zero, so the
                                   frame needs no clearing launch: the workspace's last use was a binned frame of the SAME
                                   (objects, width, height) whose render stage ran without SRH_STAGE_KEEP_BINS (the render
                                   kernel leaves every counter it read at zero), and nothing else wrote to it since.
                                   0 = unknown (a fresh or re-purposed workspace): the library clears them first.  Wrong
                                   claims cannot make the kernels leave the workspace (every list access is bounded by
                                   the list's capacity) but give a wrong image. */
  int32_t per_view;             /* srh_render_views only: SRH_VIEWS_* bits -- which of `objects`, `lights`, `materials` point
                                   to ARRAYS of n_views structs (one scene per view) instead of one struct for all views.
                                   Every view's objects must have the same batches (types and counts) as view 0's. */
  int32_t reserved1;
} SrhParams;

/* SrhParams.per_view: the reference's real batch loop changes geometry, eye and light per element
 * (diffrend/torch/GAN/gan.py:325-378: disk.pos / disk.normal = samples[idx], camera eye, lights.pos[0]) */
enum { SRH_VIEWS_OBJECTS = 1, SRH_VIEWS_LIGHTS = 2, SRH_VIEWS_MATERIALS = 4 };

/* SrhParams.stages.  SRH_STAGE_KEEP_BINS (with SRH_STAGE_RENDER): the render kernel leaves the bin counters as they are,
 * so the same bins can be rendered again (measurement builds); the workspace is then NOT clean afterwards. */
enum { SRH_STAGE_BIN = 1, SRH_STAGE_RENDER = 2, SRH_STAGE_KEEP_BINS = 4 };

int srh_abi_version(void);
const char* srh_last_error(void);

/* bytes of caller-provided device scratch needed to render `objects` at up to width x height pixels
 * (per-frame primitive records and tile bins).  Returns 0 and sets srh_last_error on invalid input.
 * The buffer must be 256-byte aligned. */
size_t srh_workspace_bytes(const SrhObjects* objects, int32_t width, int32_t height);

/* replaces generate_rays (numpy/renderer.py:145-169): unit ray directions for rows [row0,row1),
 * written as the reference returns them, ray_dir (4, n) row-major with n = (row1-row0)*W. */
int srh_generate_rays(const SrhCamera* camera, int32_t row0, int32_t row1, float* ray_dir, void* stream);

/* replaces render() (numpy/renderer.py:204-272): rays -> all-pairs intersection -> nearest valid hit ->
 * Lambert shading over the point lights -> clip -> tonemap.
 *   image   (rows, W, 3) float32
 *   depth   (rows, W)    float32, +inf where nothing is hit (:228)
 *   nearest (rows, W)    int32 global primitive index, 0 where nothing is hit (:223); may be NULL */
int srh_render_fwd(const SrhCamera* camera, const SrhObjects* objects, const SrhLights* lights,
                   const SrhMaterials* materials, const SrhParams* params,
                   void* workspace, size_t workspace_bytes,
                   float* image, float* depth, int32_t* nearest, void* stream);

/* Gradient accumulators (device pointers; NULL = gradient not wanted).  srh_render_bwd ADDS into them with fp32
 * atomics, so the caller zero-fills them first; sums over many pixels are therefore reproducible only to
 * rounding.  Layouts equal the corresponding inputs. */
typedef struct SrhGrads {
  float* pos[SRH_MAX_SEGMENTS];      /* (count,4)   disk, plane, sphere; w gets no gradient */
  float* normal[SRH_MAX_SEGMENTS];   /* (count,4)   disk, plane, triangle (the un-normalised input normal) */
  float* radius[SRH_MAX_SEGMENTS];   /* (count)     sphere; a disc's radius has no gradient (masks are piecewise constant) */
  float* face[SRH_MAX_SEGMENTS];     /* (count,3,4) triangle: only vertex 0, the plane point, is differentiated */
  float* lights_pos;                 /* (n_lights,4) */
  float* colors;                     /* (n_colors,3) */
  float* albedo;                     /* (n_materials,3) */
  float* coeffs;                     /* (n_materials,3) SRH_SHADING_TORCH only: material coefficients (c0, c1, c2) */
  float* attenuation;                /* (n_lights,3)    SRH_SHADING_TORCH only: (kc, kl, kq) */
  float* ambient;                    /* (3)             SRH_SHADING_TORCH only */
} SrhGrads;

/* Analytic backward of srh_render_fwd: the vector-Jacobian product of (image, depth) w.r.t. the scene arrays for the
 * upstream gradients grad_image (rows,W,3) and grad_depth (rows,W; may be NULL), using the winners saved by the
 * forward pass (`nearest`, and `depth` to tell hit pixels from background).  Defined exactly as autograd through
 * the reference's differentiable backend defines it (diffrend/torch/renderer.py:136-355, torch/utils.py:238-366):
 * the nearest-hit selection and all masks are piecewise constant, so gradients flow only through the winner's hit
 * distance, hit point, normal, albedo and the lights.  With SRH_SHADING_TORCH the shading model differentiated is that
 * backend's Phong model (attenuation, specular coefficients, ambient; relus, the double_sided sign and the masks are
 * constants), and coeffs / attenuation / ambient gradients are available.  Row strides and the row range come from
 * `params` as in the forward call. */
int srh_render_bwd(const SrhCamera* camera, const SrhObjects* objects, const SrhLights* lights,
                   const SrhMaterials* materials, const SrhParams* params,
                   void* workspace, size_t workspace_bytes,
                   const float* grad_image, const float* grad_depth,
                   const int32_t* nearest, const float* depth,
                   const SrhGrads* grads, void* stream);

/* Many views in one call: the batch axis of the reference's real callers, which render one view per
 * render() call in a Python loop (diffrend/torch/GAN/gan.py:325-378, torch/batch_render.py:36-53).  `cameras` is an
 * array of n_views cameras with one viewport size; objects / lights / materials are one scene for all views or, per
 * SrhParams.per_view, one per view (same batch structure); `params` is shared (binned mode; row range and output row strides as
 * in srh_render_fwd, so a multi-GPU rank can render its slab of a whole batch of frames); the outputs are stacked:
 * view v starts v * rows * row_stride elements after view 0 in images (n_views,rows,W,3), depths (n_views,rows,W)
 * and nearests (may be NULL).  Every kernel of the
 * frame pipeline is launched once for the whole batch (the view is a grid dimension), so small views cost neither
 * three launches each nor an idle GPU.  Results equal srh_render_fwd per view.  The workspace must hold
 * srh_workspace_bytes_views(...) bytes.  Calls on one device are serialised on that device's staging ring. */
size_t srh_workspace_bytes_views(const SrhObjects* objects, int32_t width, int32_t height, int32_t n_views);
int srh_render_views(int32_t n_views, const SrhCamera* cameras, const SrhObjects* objects, const SrhLights* lights,
                     const SrhMaterials* materials, const SrhParams* params, void* workspace, size_t workspace_bytes,
                     float* images, float* depths, int32_t* nearests, void* stream);

/* The torch backend's `shadow=True` (diffrend/torch/renderer.py:291-314) as a second pass over a frame rendered by
 * srh_render_fwd with the same camera / scene / params (SRH_SHADING_TORCH): per hit pixel and light a shadow ray from
 * the fragment (started 0.1 towards the light) against every primitive, all pairs in fp64; a light is visible unless
 * a primitive other than the fragment's own is hit before it.  `image` (rows,W,3) is overwritten with the re-shaded
 * frame; `visibility` (rows,W) uint64, bit l = light l visible, may be NULL.
 * With a workspace of srh_shadow_workspace_bytes(...) bytes the candidates of a shadow ray come from tile bins built
 * in each light's own screen space (every candidate still goes through the same fp64 test, so the result equals the
 * all-pairs pass bit for bit); with the smaller srh_workspace_bytes(...) workspace, or params->mode =
 * SRH_MODE_EXACT, the pass is the reference's O(pixels x lights x primitives) loop. */
size_t srh_shadow_workspace_bytes(const SrhObjects* objects, int32_t width, int32_t height, int32_t n_lights);
int srh_shadow_shade(const SrhCamera* camera, const SrhObjects* objects, const SrhLights* lights,
                     const SrhMaterials* materials, const SrhParams* params, void* workspace, size_t workspace_bytes,
                     const int32_t* nearest, const float* depth, float* image, uint64_t* visibility, void* stream);

/* Measurement helper: where a binned frame's list lengths live in its workspace, so that a caller can count the
 * (pixel, primitive) tests a frame really executes (bench.py: executed_pair_tests) or weigh image rows by their work
 * (surf_renderer_amd.dist.cost_weighted_slabs).  After an SRH_STAGE_BIN call for rows [row0, row1), the 32-bit words
 * at workspace + *offset_bytes hold:  [4 * set + s] length of batch s's frame-wide list, with set = word [9];
 * [64 + s * *ntiles_pad + ty * *tiles_x + tx] length of the bin of batch s and tile (tx, ty) -- lists longer than
 * *bin_cap are cut there (their primitives are on the frame-wide list instead).  The render stage zeroes them again. */
int srh_bin_counters(const SrhObjects* objects, int32_t width, int32_t height, int32_t row0, int32_t row1,
                     size_t* offset_bytes, int32_t* tiles_x, int32_t* tiles_y, int32_t* ntiles_pad, int32_t* bin_cap);

/* measurement helpers: timing-enabled HIP events usable as SrhParams.ev_start / ev_stop.
 * srh_event_elapsed_ms waits for `stop` to complete (the only call here that blocks the host). */
int srh_event_create(void** event);
int srh_event_destroy(void* event);
int srh_event_elapsed_ms(void* start, void* stop, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* SRH_H */

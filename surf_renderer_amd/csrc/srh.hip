// libsrh.so -- MI355X (gfx950) render(scene) backend: kernels + the C ABI declared in include/srh.h.
//
// Launch structure of one frame (all on the caller's stream, no host sync):
//   k_prep        one thread per primitive: per-frame records (unit normal, plane offset, eye-relative
//                 centre, ...) in fp64, plus the fp32 reject records of the FAST mode
//   k_render_*    one 256-thread workgroup per 64x4-pixel tile; every wave owns 64 consecutive pixels
//                 of one image row, so depth / nearest / RGB stores are full-wave coalesced rows
// The primitive stream is wave-uniform (every lane walks the same record), so records arrive through
// the scalar cache into SGPRs; no LDS staging is needed for uniform reads.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "srh.h"
#include "srh_device.h"

using namespace srh;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

constexpr size_t kAlign = 256;
size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

constexpr int kRec32Stride[4] = {8, 4, 8, 12};  // floats per primitive reject record (FAST mode)

// ------------------------------------------------------------------------------------------------
// k_prep: per-frame primitive records
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep(SegDev S, double ox, double oy, double oz, double* rec64) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S.count) return;
  const double o[3] = {ox, oy, oz};
  double* R = rec64 + (size_t)i * kRec64Stride[S.type];

  double nh[3] = {0, 0, 0};
  if (S.type != SRH_PRIM_SPHERE) {
    // ops.normalize: divide by the 4-D length, by 1 if that is zero (numpy/ops.py:18-26)
    const float* q = S.normal + 4 * (size_t)i;
    const double v[4] = {(double)q[0], (double)q[1], (double)q[2], (double)q[3]};
    double len = sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + v[3] * v[3]);
    if (!(fabs(len) > 0.0)) len = 1.0;
    nh[0] = v[0] / len; nh[1] = v[1] / len; nh[2] = v[2] / len;
  }
  if (S.type == SRH_PRIM_SPHERE) {
    const float* c = S.pos + 4 * (size_t)i;
    const double r = (double)S.radius[i];
    const double oc[3] = {o[0] - (double)c[0], o[1] - (double)c[1], o[2] - (double)c[2]};
    R[0] = oc[0]; R[1] = oc[1]; R[2] = oc[2];
    R[3] = ((oc[0] * oc[0] + oc[1] * oc[1]) + oc[2] * oc[2]) - r * r;     // numpy/renderer.py:22
    return;
  }
  // point on the plane: pos, or vertex 0 of the triangle (numpy/renderer.py:107)
  const float* pp = (S.type == SRH_PRIM_TRIANGLE) ? S.face + 12 * (size_t)i : S.pos + 4 * (size_t)i;
  const double p[3] = {(double)pp[0], (double)pp[1], (double)pp[2]};
  // dist - n^.eye (numpy/renderer.py:62,69)
  const double dist = (p[0] * nh[0] + p[1] * nh[1]) + p[2] * nh[2];
  const double neye = (nh[0] * o[0] + nh[1] * o[1]) + nh[2] * o[2];
  R[0] = nh[0]; R[1] = nh[1]; R[2] = nh[2];
  R[3] = dist - neye;
  if (S.type == SRH_PRIM_DISK) {
    const double r = (double)S.radius[i];
    R[4] = o[0] - p[0]; R[5] = o[1] - p[1]; R[6] = o[2] - p[2];
    R[7] = r * r;
  } else if (S.type == SRH_PRIM_TRIANGLE) {
    const float* f = S.face + 12 * (size_t)i;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int w = (v + 1) % 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        R[4 + 3 * v + k] = (double)f[4 * v + k];
        R[13 + 3 * v + k] = (double)f[4 * w + k] - (double)f[4 * v + k];
      }
    }
    R[22] = 0.0; R[23] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------
// k_rays: generate_rays as an output (reference returns 'ray_dir' (4,N))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rays(FrameDev F, float* ray_dir) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int r = F.row0 + blockIdx.y * 4 + threadIdx.y;
  if (c >= F.W || r >= F.row1) return;
  double d[3];
  pixel_ray(F, c, r, d);
  const size_t n = (size_t)(F.row1 - F.row0) * F.W;
  const size_t p = (size_t)(r - F.row0) * F.W + c;
  ray_dir[p] = (float)d[0];
  ray_dir[n + p] = (float)d[1];
  ray_dir[2 * n + p] = (float)d[2];
  ray_dir[3 * n + p] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// k_render_exact: every (pixel, primitive) pair through the fp64 intersection
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_pixel(const FrameDev& F, int c, int r, const float rgb[3], double z, int win,
                                            float* __restrict__ image, float* __restrict__ depth,
                                            int32_t* __restrict__ nearest) {
  const size_t row = (size_t)(r - F.row0);
  float* px = image + row * F.img_stride + 3 * (size_t)c;
  px[0] = rgb[0];
  px[1] = rgb[1];
  px[2] = rgb[2];
  depth[row * F.depth_stride + c] = (float)z;
  if (nearest) nearest[row * F.near_stride + c] = win;
}

__global__ __launch_bounds__(256) void k_render_exact(FrameDev F, float* __restrict__ image,
                                                       float* __restrict__ depth, int32_t* __restrict__ nearest) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int r = F.row0 + blockIdx.y * 4 + threadIdx.y;
  const bool live = (c < F.W) && (r < F.row1);
  double d[3];
  pixel_ray(F, live ? c : F.W - 1, live ? r : F.row1 - 1, d);

  double best = __builtin_inf();
  int besti = 0;
  for (int s = 0; s < F.nseg; ++s) {
    const SegDev& S = F.seg[s];
    const int stride = kRec64Stride[S.type];
    for (int i = 0; i < S.count; ++i) {
      const double t = hit_any64(S.type, S.rec64 + (size_t)i * stride, F.o, d);
      resolve(F, t, S.first + i, best, besti);
    }
  }
  float rgb[3];
  shade_pixel(F, d, best, besti, rgb);
  if (live) store_pixel(F, c, r, rgb, best, besti, image, depth, nearest);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct WsLayout {
  size_t off64[SRH_MAX_SEGMENTS];
  size_t off32[SRH_MAX_SEGMENTS];
  size_t total;
};

int check_objects(const SrhObjects* ob) {
  if (!ob) return fail(SRH_E_NULL, "objects is NULL");
  if (ob->n_segments < 1 || ob->n_segments > SRH_MAX_SEGMENTS)
    return fail(SRH_E_RANGE, "n_segments = %d, expected 1..%d", ob->n_segments, SRH_MAX_SEGMENTS);
  long long total = 0;
  for (int s = 0; s < ob->n_segments; ++s) {
    const SrhSegment& g = ob->seg[s];
    if (g.type < 0 || g.type > 3) return fail(SRH_E_TYPE, "segment %d: unknown primitive type %d", s, g.type);
    if (g.count < 1) return fail(SRH_E_RANGE, "segment %d: count = %d (empty batches are not allowed)", s, g.count);
    if (!g.material_idx) return fail(SRH_E_NULL, "segment %d: material_idx is NULL", s);
    const bool need_pos = g.type != SRH_PRIM_TRIANGLE, need_nrm = g.type != SRH_PRIM_SPHERE;
    const bool need_rad = g.type == SRH_PRIM_DISK || g.type == SRH_PRIM_SPHERE;
    if (need_pos && !g.pos) return fail(SRH_E_NULL, "segment %d: pos is NULL", s);
    if (need_nrm && !g.normal) return fail(SRH_E_NULL, "segment %d: normal is NULL", s);
    if (need_rad && !g.radius) return fail(SRH_E_NULL, "segment %d: radius is NULL", s);
    if (g.type == SRH_PRIM_TRIANGLE && !g.face) return fail(SRH_E_NULL, "segment %d: face is NULL", s);
    total += g.count;
  }
  if (total > 0x7fffffffLL) return fail(SRH_E_RANGE, "too many primitives (%lld)", total);
  return SRH_OK;
}

WsLayout layout_for(const SrhObjects* ob) {
  WsLayout L;
  size_t off = 0;
  for (int s = 0; s < ob->n_segments; ++s) {
    const SrhSegment& g = ob->seg[s];
    L.off64[s] = off;
    off = align_up(off + (size_t)g.count * kRec64Stride[g.type] * sizeof(double));
    L.off32[s] = off;
    off = align_up(off + (size_t)g.count * kRec32Stride[g.type] * sizeof(float));
  }
  L.total = off;
  return L;
}

// numpy/renderer.py:145-163 + numpy/ops.py:88-115 on the host, in fp64, same operation order.
int camera_to_frame(const SrhCamera* cam, FrameDev* F) {
  if (!cam) return fail(SRH_E_NULL, "camera is NULL");
  const int W = cam->viewport[2] - cam->viewport[0], H = cam->viewport[3] - cam->viewport[1];
  if (W < 1 || H < 1) return fail(SRH_E_RANGE, "empty viewport %d x %d", W, H);
  if (cam->eye[3] != 1.0) return fail(SRH_E_CAMERA, "camera.eye must have w == 1");
  if (cam->up[3] != 0.0) return fail(SRH_E_CAMERA, "camera.up must have w == 0");
  double z[4], zl = 0;
  for (int i = 0; i < 4; ++i) { z[i] = cam->eye[i] - cam->at[i]; zl += z[i] * z[i]; }
  zl = sqrt(zl);
  double ul = sqrt(cam->up[0] * cam->up[0] + cam->up[1] * cam->up[1] + cam->up[2] * cam->up[2]);
  if (!(zl > 0) || !(ul > 0)) return fail(SRH_E_CAMERA, "degenerate camera: eye == at or up == 0");
  double y[3];
  for (int i = 0; i < 3; ++i) { z[i] /= zl; y[i] = cam->up[i] / ul; }
  const double x[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
  const double h = tan(cam->fovy / 2) * 2 * cam->focal_length;
  const double w = h * ((double)W / (double)H);
  for (int i = 0; i < 3; ++i) { F->o[i] = cam->eye[i]; F->bx[i] = x[i]; F->by[i] = y[i]; F->bz[i] = z[i]; }
  F->half_w = w / 2;
  F->half_h = h / 2;
  F->focal = cam->focal_length;
  F->step_x = W > 1 ? 2.0 / (W - 1) : 0.0;
  F->step_y = H > 1 ? -2.0 / (H - 1) : 0.0;
  F->near_clip = cam->near_clip;
  F->far_clip = cam->far_clip;
  F->W = W;
  F->H = H;
  return SRH_OK;
}

int check_rows(const FrameDev& F, int row0, int row1) {
  if (row0 < 0 || row1 > F.H || row0 >= row1)
    return fail(SRH_E_RANGE, "row range [%d,%d) outside the %d image rows", row0, row1, F.H);
  return SRH_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int srh_abi_version(void) { return SRH_ABI_VERSION; }

const char* srh_last_error(void) { return g_err; }

size_t srh_workspace_bytes(const SrhObjects* objects) {
  if (check_objects(objects) != SRH_OK) return 0;
  return layout_for(objects).total;
}

int srh_generate_rays(const SrhCamera* camera, int32_t row0, int32_t row1, float* ray_dir, void* stream) {
  FrameDev F;
  memset(&F, 0, sizeof(F));
  int rc = camera_to_frame(camera, &F);
  if (rc) return rc;
  if ((rc = check_rows(F, row0, row1))) return rc;
  if (!ray_dir) return fail(SRH_E_NULL, "ray_dir is NULL");
  F.row0 = row0;
  F.row1 = row1;
  const dim3 block(64, 4), grid((F.W + 63) / 64, (row1 - row0 + 3) / 4);
  hipLaunchKernelGGL(k_rays, grid, block, 0, (hipStream_t)stream, F, ray_dir);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SRH_OK : hip_fail(e, "k_rays launch");
}

int srh_render_fwd(const SrhCamera* camera, const SrhObjects* objects, const SrhLights* lights,
                   const SrhMaterials* materials, const SrhParams* params, void* workspace,
                   size_t workspace_bytes, float* image, float* depth, int32_t* nearest, void* stream) {
  FrameDev F;
  memset(&F, 0, sizeof(F));
  int rc = camera_to_frame(camera, &F);
  if (rc) return rc;
  if ((rc = check_objects(objects))) return rc;
  if (!lights || !materials || !params) return fail(SRH_E_NULL, "lights / materials / params is NULL");
  if ((rc = check_rows(F, params->row0, params->row1))) return rc;
  if (!image || !depth) return fail(SRH_E_NULL, "image / depth is NULL");
  if (lights->n_lights < 0 || lights->n_lights > SRH_MAX_LIGHTS)
    return fail(SRH_E_RANGE, "n_lights = %d, expected 0..%d", lights->n_lights, SRH_MAX_LIGHTS);
  if (lights->n_lights > 0 && (!lights->pos || !lights->color_idx || !lights->colors || lights->n_colors < 1))
    return fail(SRH_E_NULL, "lights arrays missing");
  if (materials->n_materials < 1 || !materials->albedo) return fail(SRH_E_NULL, "materials.albedo missing");
  if (params->mode < SRH_MODE_AUTO || params->mode > SRH_MODE_FAST) return fail(SRH_E_TYPE, "unknown mode %d", params->mode);
  const WsLayout L = layout_for(objects);
  if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace % kAlign) != 0)
    return fail(SRH_E_WORKSPACE, "workspace: need %zu bytes, 256-byte aligned (got %zu at %p)", L.total,
                workspace_bytes, workspace);

  F.row0 = params->row0;
  F.row1 = params->row1;
  F.gamma = params->gamma;
  F.tonemap = params->tonemap_gamma ? 1 : 0;
  F.img_stride = params->image_row_stride ? params->image_row_stride : 3 * (int64_t)F.W;
  F.depth_stride = params->depth_row_stride ? params->depth_row_stride : (int64_t)F.W;
  F.near_stride = params->nearest_row_stride ? params->nearest_row_stride : (int64_t)F.W;
  if (F.img_stride < 3 * (int64_t)F.W || F.depth_stride < F.W || F.near_stride < F.W)
    return fail(SRH_E_RANGE, "output row strides shorter than a row");
  F.nseg = objects->n_segments;
  F.nlights = lights->n_lights;
  F.ncolors = lights->n_colors;
  F.nmat = materials->n_materials;
  F.lpos = lights->pos;
  F.lcidx = lights->color_idx;
  F.colors = lights->colors;
  F.albedo = materials->albedo;
  int first = 0;
  for (int s = 0; s < F.nseg; ++s) {
    const SrhSegment& g = objects->seg[s];
    SegDev& S = F.seg[s];
    S.type = g.type;
    S.count = g.count;
    S.first = first;
    S.rec64 = (const double*)((char*)workspace + L.off64[s]);
    S.rec32 = (const float*)((char*)workspace + L.off32[s]);
    S.pos = g.pos;
    S.normal = g.normal;
    S.radius = g.radius;
    S.face = g.face;
    S.mat = g.material_idx;
    first += g.count;
  }
  F.total = first;

  hipStream_t st = (hipStream_t)stream;
  for (int s = 0; s < F.nseg; ++s) {
    const SegDev& S = F.seg[s];
    hipLaunchKernelGGL(k_prep, dim3((S.count + 255) / 256), dim3(256), 0, st, S, F.o[0], F.o[1], F.o[2],
                       (double*)S.rec64);
  }
  const dim3 block(64, 4), grid((F.W + 63) / 64, (F.row1 - F.row0 + 3) / 4);
  if (params->ev_start) hipEventRecord((hipEvent_t)params->ev_start, st);
  hipLaunchKernelGGL(k_render_exact, grid, block, 0, st, F, image, depth, nearest);
  if (params->ev_stop) hipEventRecord((hipEvent_t)params->ev_stop, st);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SRH_OK : hip_fail(e, "render launch");
}

int srh_event_create(void** event) {
  if (!event) return fail(SRH_E_NULL, "event is NULL");
  hipEvent_t ev;
  hipError_t e = hipEventCreate(&ev);
  if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
  *event = (void*)ev;
  return SRH_OK;
}

int srh_event_destroy(void* event) {
  if (!event) return SRH_OK;
  hipError_t e = hipEventDestroy((hipEvent_t)event);
  return e == hipSuccess ? SRH_OK : hip_fail(e, "hipEventDestroy");
}

int srh_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return fail(SRH_E_NULL, "event / ms is NULL");
  hipError_t e = hipEventSynchronize((hipEvent_t)stop);
  if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
  e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
  return e == hipSuccess ? SRH_OK : hip_fail(e, "hipEventElapsedTime");
}

}  // extern "C"

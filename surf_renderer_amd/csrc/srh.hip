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

#include <mutex>

#include "srh.h"
#include "srh_device.h"
#include "srh_reject.h"
#include "srh_binned.h"
#include "srh_backward.h"
#include "srh_shadow.h"

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

// ------------------------------------------------------------------------------------------------
// k_prep: per-frame primitive records
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void prep_record64(const SegDev& S, int type, int i, const double o[3], bool tch, double* R);

// Can no pixel of rows [row0, row1) see anything of the ball (centre x relative to the eye, radius rho)?  The ball's
// points have a in a0 +- rho |m_a| and g in g0 +- rho |m_g| (Cauchy-Schwarz) and lie on image row g / a; rays exist
// only through integer rows, hence the half-row slack.  A ball that reaches the eye plane (a <= 0) is kept.
__device__ inline bool ball_misses_slab(const FrameDev& F, const double x[3], double rho) {
  const double a0 = dot3(F.slab_ma, x), g0 = dot3(F.slab_mg, x);
  const double a_lo = a0 - rho * F.slab_na, a_hi = a0 + rho * F.slab_na;
  const double g_lo = g0 - rho * F.slab_ng, g_hi = g0 + rho * F.slab_ng;
  // fp64 side: a0 and g0 carry 2^-52 of their absolute terms.  The cull is used only while a_lo keeps 1e-6 of them
  // (relative error of a below 2^-32, of the rows below 1e-6 of a row: inside the half-row slack); a ball that far off
  // axis, or that cancelled (centre ~1e20 away, radius to match), is simply kept
  if (!(a_lo > 1.0e-6 * (abs_dot3(F.slab_ma, x) + rho * F.slab_na)) || !isfinite(a_hi + g_lo + g_hi)) return false;
  const double r_lo = g_lo / (g_lo >= 0.0 ? a_hi : a_lo), r_hi = g_hi / (g_hi >= 0.0 ? a_lo : a_hi);
  return r_hi < (double)F.row0 - 0.5 || r_lo > (double)F.row1 - 0.5;
}

// multi-GPU row slabs: most primitives project outside a rank's rows; they are recognised from their bounding ball
// before any of the per-frame records is computed, and are simply not binned (no list refers to their records)
__device__ inline bool primitive_misses_slab(const FrameDev& F, const SegDev& S, int i) {
  double x[3], rho;
  // numpy semantics, near <= 0: a sphere whose line a ray MISSES yields the valid distance 0 (Q2) -- on every pixel
  // of the image, wherever the sphere projects; it can never be culled
  if (S.type == SRH_PRIM_SPHERE && !(F.near_clip > 0.0)) return false;
  if (S.type == SRH_PRIM_DISK || S.type == SRH_PRIM_SPHERE) {
    const float* c = S.pos + 4 * (size_t)i;
    for (int k = 0; k < 3; ++k) x[k] = (double)c[k] - F.o[k];
    rho = fabs((double)S.radius[i]);
  } else if (S.type == SRH_PRIM_TRIANGLE) {
    const float* f = S.face + 12 * (size_t)i;
    double cen[3];
    for (int k = 0; k < 3; ++k) cen[k] = ((double)f[k] + (double)f[4 + k] + (double)f[8 + k]) / 3.0;
    rho = 0.0;
    for (int v = 0; v < 3; ++v) {
      const double w[3] = {(double)f[4 * v] - cen[0], (double)f[4 * v + 1] - cen[1], (double)f[4 * v + 2] - cen[2]};
      rho = fmax(rho, sqrt(dot3(w, w)));
    }
    rho *= 1.0000001;
    for (int k = 0; k < 3; ++k) x[k] = cen[k] - F.o[k];
  } else {
    return false;
  }
  return ball_misses_slab(F, x, rho);
}

// TYPE = the batch's primitive type (the host launches the matching instantiation): the fp64 record and the reject
// record are built in REGISTERS, stored once, and the tile box and the bin placement work from the register copies --
// a thread never waits for its own stores to come back (measured: the kernel was 76 % s_waitcnt).
template <int TYPE>
__device__ __forceinline__ void prep_body(const FrameDev& F, int s, double* rec64, float* rec32) {
  const SegDev& S = F.seg[s];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S.count) return;
  // which set of frame-wide list lengths this frame counts into (srh_device.h: kLargeNext); stable for the whole binning
  const uint32_t set = F.tilerange ? (F.counters[kLargeNext] & 1u) : 0u;
  if (s == 0 && i == 0) {
    if (F.tilerange) F.counters[kLargeNow] = set;          // ... and the render kernel reads the same one
    // per-frame fp64 copy of the lights for the fragment stage: position, colour looked up through color_idx
    double* L = const_cast<double*>(F.lights64);
    for (int l = 0; l < F.nlights; ++l) {
      const int ci = clampi(F.lcidx[l], 0, F.ncolors - 1);
      for (int k = 0; k < 3; ++k) {
        L[6 * l + k] = (double)F.lpos[4 * l + k];
        L[6 * l + 3 + k] = (double)F.colors[3 * ci + k];
      }
    }
  }
  if (F.tilerange && F.slab_cull && primitive_misses_slab(F, S, i)) {
    uint16_t* tr = F.tilerange + 4 * (size_t)(S.first + i);
    tr[0] = 1; tr[1] = 0; tr[2] = 0; tr[3] = 0;                   // not binned
    return;
  }
  constexpr int N64 = kRec64Stride[TYPE], N32 = kRec32Stride[TYPE];
  double R[N64];
  prep_record64(S, TYPE, i, F.o, F.shading != 0, R);
  // screen-space reject record of the FAST / binned modes, from the fp64 record
  float Q[N32];
  const PixelBasis B = pixel_basis(F);
  const bool near_pos = F.near_clip > 0.0;
  if (TYPE == SRH_PRIM_DISK) disk_reject_record(R, F.o, B, F.W, F.H, Q);
  else if (TYPE == SRH_PRIM_SPHERE) sphere_reject_record(R, B, F.W, F.H, near_pos, F.shading != 0, Q);
  else if (TYPE == SRH_PRIM_TRIANGLE) triangle_reject_record(R, F.o, B, F.W, F.H, near_pos, Q);
  else plane_reject_record(R, B, F.W, F.H, Q);
  {
    double2* r2 = reinterpret_cast<double2*>(rec64 + (size_t)i * N64);      // records are 16-byte aligned (strides 4, 8, 24)
#pragma unroll
    for (int k = 0; k < N64 / 2; ++k) r2[k] = make_double2(R[2 * k], R[2 * k + 1]);
    float4* q4 = reinterpret_cast<float4*>(rec32 + (size_t)i * N32);
#pragma unroll
    for (int k = 0; k < N32 / 4; ++k) q4[k] = make_float4(Q[4 * k], Q[4 * k + 1], Q[4 * k + 2], Q[4 * k + 3]);
  }
  if (F.tilerange) {
    // light views: a primitive that comes within near_ball of the eye can block a shadow ray from BEHIND the light
    // (the reference accepts hits up to 0.1 beyond it); its screen-space shape says nothing about that, so every
    // query tests it
    bool near_eye = false;
    if (F.near_ball > 0.0) {
      // distance from the light to the nearest point the primitive can have, as a difference of two lengths -- which
      // cancels for a primitive as large as it is far (centre 1e20 away, radius 1e20): 2^-46 of the lengths' sum,
      // 64x their rounding, comes off
      double dmin = 0.0, mag = 0.0;
      if (TYPE == SRH_PRIM_DISK) {
        const double oc[3] = {F.o[0] - R[4], F.o[1] - R[5], F.o[2] - R[6]};
        const double dc = sqrt(dot3(oc, oc)), rr = sqrt(fabs(R[7]));
        dmin = dc - rr; mag = dc + rr;
      }
      else if (TYPE == SRH_PRIM_SPHERE) {
        // (the radius itself: |oc|^2 - (|oc|^2 - r^2) gives r^2 back only to 2^-52 |oc|^2)
        const double dc = sqrt(dot3(R, R)), rr = fabs((double)S.radius[i]);
        dmin = dc - rr; mag = dc + rr;
      }
      else if (TYPE == SRH_PRIM_TRIANGLE) {
        double far2 = 0.0, near2 = 1.0e300;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const double w[3] = {R[4 + 3 * v] - F.o[0], R[5 + 3 * v] - F.o[1], R[6 + 3 * v] - F.o[2]};
          near2 = fmin(near2, dot3(w, w));
          const double e[3] = {R[13 + 3 * v], R[14 + 3 * v], R[15 + 3 * v]};
          far2 = fmax(far2, dot3(e, e));
        }
        dmin = sqrt(near2) - sqrt(far2);                        // every point is within one edge length of a vertex
        mag = sqrt(near2) + sqrt(far2);
      }
      dmin -= 1.4210854715202004e-14 * mag;
      near_eye = !(dmin > F.near_ball);                         // NaN -> large
      // what the shadow pass skips candidates by: no point of the primitive is closer to the light than this (rounded
      // DOWN to fp32; 0 = unknown: planes, non-finite geometry)
      const float nd = (TYPE != SRH_PRIM_PLANE && dmin > 0.0 && dmin < 1.0e30) ? (float)dmin * 0.999999f : 0.0f;
      F.neardist[S.first + i] = nd;
    }
    const TileBox box = bin_primitive(F, s, TYPE, Q, S.first + i, set, near_eye);
#if SRH_FUSE_BIN
    if (box.tx0 <= box.tx1) bin_place<1>(F, s, TYPE, S.first, Q, S.first + i, 0, box.tx0, box.ty0, box.tx1, box.ty1, set);
#else
    (void)box;
#endif
  }
}

// Four waves per SIMD (at most 128 VGPRs), asked for explicitly: left alone hipcc takes what it likes -- 150 registers for
// the disc instantiation once the fp64 trust terms of srh_reject.h went in, three waves per SIMD -- and the prep waves of
// the frames in flight then hold their slots longer beside the render waves: config 5 went from 0.078 to 0.093 ms per
// frame on that alone.  At 128 the compiler needs no spills.
#ifndef SRH_PREP_WAVES
#define SRH_PREP_WAVES 4
#endif
#define SRH_PREP_ATTR __attribute__((amdgpu_waves_per_eu(SRH_PREP_WAVES)))
template <int TYPE>
__global__ __launch_bounds__(kBinBlock) SRH_PREP_ATTR void k_prep(FrameDev F, int s, double* rec64, float* rec32) {
  // the frame's constants into the workspace, where the render kernel reads them (FrameDev::self; srh_binned.h)
  if (F.self && s == 0 && blockIdx.x == 0) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&F);
    uint32_t* dst = reinterpret_cast<uint32_t*>(F.self);
    for (unsigned i = threadIdx.x; i < sizeof(FrameDev) / 4; i += kBinBlock) dst[i] = src[i];
  }
  prep_body<TYPE>(F, s, rec64, rec32);
}

template <int TYPE>
__global__ __launch_bounds__(kBinBlock) SRH_PREP_ATTR void k_prep_views(const FrameDev* __restrict__ Fs, int s) {
  const FrameDev& F = Fs[blockIdx.y];
  prep_body<TYPE>(F, s, const_cast<double*>(F.seg[s].rec64), const_cast<float*>(F.seg[s].rec32));
}

// host side: the instantiation for the batch's type
static void launch_prep(const FrameDev& F, int s, hipStream_t st) {
  const SegDev& S = F.seg[s];
  const dim3 grid((S.count + kBinBlock - 1) / kBinBlock), block(kBinBlock);
  double* r64 = (double*)S.rec64;
  float* r32 = (float*)S.rec32;
  switch (S.type) {
    case SRH_PRIM_DISK: hipLaunchKernelGGL(k_prep<SRH_PRIM_DISK>, grid, block, 0, st, F, s, r64, r32); break;
    case SRH_PRIM_PLANE: hipLaunchKernelGGL(k_prep<SRH_PRIM_PLANE>, grid, block, 0, st, F, s, r64, r32); break;
    case SRH_PRIM_SPHERE: hipLaunchKernelGGL(k_prep<SRH_PRIM_SPHERE>, grid, block, 0, st, F, s, r64, r32); break;
    default: hipLaunchKernelGGL(k_prep<SRH_PRIM_TRIANGLE>, grid, block, 0, st, F, s, r64, r32); break;
  }
}
static void launch_prep_views(const FrameDev& F0, const FrameDev* Fs, int s, int V, hipStream_t st) {
  const dim3 grid((F0.seg[s].count + kBinBlock - 1) / kBinBlock, V), block(kBinBlock);
  switch (F0.seg[s].type) {
    case SRH_PRIM_DISK: hipLaunchKernelGGL(k_prep_views<SRH_PRIM_DISK>, grid, block, 0, st, Fs, s); break;
    case SRH_PRIM_PLANE: hipLaunchKernelGGL(k_prep_views<SRH_PRIM_PLANE>, grid, block, 0, st, Fs, s); break;
    case SRH_PRIM_SPHERE: hipLaunchKernelGGL(k_prep_views<SRH_PRIM_SPHERE>, grid, block, 0, st, Fs, s); break;
    default: hipLaunchKernelGGL(k_prep_views<SRH_PRIM_TRIANGLE>, grid, block, 0, st, Fs, s); break;
  }
}

__device__ __forceinline__ void prep_record64(const SegDev& S, int type, int i, const double o[3], bool tch, double* R) {
  double nh[3] = {0, 0, 0};
  if (type != SRH_PRIM_SPHERE) {
    // ops.normalize: divide by the 4-D length, by 1 if that is zero (numpy/ops.py:18-26)
    const float* q = S.normal + 4 * (size_t)i;
    const double v[4] = {(double)q[0], (double)q[1], (double)q[2], (double)q[3]};
    double len = sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + v[3] * v[3]);
    // the torch backend normalises xyz only, with an eps inside the sum (torch/utils.py:131-135, :289)
    if (tch) len = sqrt(((v[0] * v[0] + 1e-10) + (v[1] * v[1] + 1e-10)) + (v[2] * v[2] + 1e-10));
    if (!(fabs(len) > 0.0)) len = 1.0;
    nh[0] = v[0] / len; nh[1] = v[1] / len; nh[2] = v[2] / len;
  }
  if (type == SRH_PRIM_SPHERE) {
    const float* c = S.pos + 4 * (size_t)i;
    const double r = (double)S.radius[i];
    const double oc[3] = {o[0] - (double)c[0], o[1] - (double)c[1], o[2] - (double)c[2]};
    R[0] = oc[0]; R[1] = oc[1]; R[2] = oc[2];
    R[3] = ((oc[0] * oc[0] + oc[1] * oc[1]) + oc[2] * oc[2]) - r * r;     // numpy/renderer.py:22
    return;
  }
  // point on the plane: pos, or vertex 0 of the triangle (numpy/renderer.py:107)
  const float* pp = (type == SRH_PRIM_TRIANGLE) ? S.face + 12 * (size_t)i : S.pos + 4 * (size_t)i;
  const double p[3] = {(double)pp[0], (double)pp[1], (double)pp[2]};
  // dist - n^.eye (numpy/renderer.py:62,69)
  const double dist = (p[0] * nh[0] + p[1] * nh[1]) + p[2] * nh[2];
  const double neye = (nh[0] * o[0] + nh[1] * o[1]) + nh[2] * o[2];
  R[0] = nh[0]; R[1] = nh[1]; R[2] = nh[2];
  R[3] = dist - neye;
  if (type == SRH_PRIM_DISK) {
    const double r = (double)S.radius[i];
    R[4] = p[0]; R[5] = p[1]; R[6] = p[2];
    R[7] = r * r;
  } else if (type == SRH_PRIM_TRIANGLE) {
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
                                            int32_t* __restrict__ nearest, const float* aux = nullptr) {
  const size_t row = (size_t)(r - F.row0);
  float* px = image + row * F.img_stride + 3 * (size_t)c;
  px[0] = rgb[0];
  px[1] = rgb[1];
  px[2] = rgb[2];
  depth[row * F.depth_stride + c] = background_depth(F, z);
  if (nearest) nearest[row * F.near_stride + c] = win;
  if (aux) store_aux(F, row, c, aux);
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
      const double t = hit_any64(S.type, S.rec64 + (size_t)i * stride, F.o, d, F.shading != 0);
      resolve(F, t, S.first + i, best, besti);
    }
  }
  float rgb[3], aux[6];
  const bool want_aux = F.normal_out || F.pos_out;
  shade_pixel(F, d, best, besti, rgb, want_aux ? aux : nullptr);
  if (live) store_pixel(F, c, r, rgb, best, besti, image, depth, nearest, want_aux ? aux : nullptr);
}

// ------------------------------------------------------------------------------------------------
// k_render_ortho: orthographic projection of the torch backend (torch/utils.py:461-468): every ray has the direction
// -z of the camera basis and its own origin eye + x X + y Y.  All pairs in fp64 (the screen-space reject records are
// derived for a pinhole); the reference's own ortho branch only works for images below one 4096-pixel tile.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_render_ortho(FrameDev F, float* __restrict__ image,
                                                       float* __restrict__ depth, int32_t* __restrict__ nearest) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int r = F.row0 + blockIdx.y * 4 + threadIdx.y;
  const bool live = (c < F.W) && (r < F.row1);
  const int cc = live ? c : F.W - 1, rr = live ? r : F.row1 - 1;
  const double xs = (F.W > 1 && cc == F.W - 1) ? 1.0 : (cc * F.step_x + -1.0);
  const double ys = (F.H > 1 && rr == F.H - 1) ? -1.0 : (rr * F.step_y + 1.0);
  const double X = xs * F.half_w, Y = ys * F.half_h;
  double q[3], org[3], d[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    q[i] = F.bx[i] * X + F.by[i] * Y;
    org[i] = F.o[i] + q[i];
    d[i] = -F.bz[i];
  }
  double best = __builtin_inf();
  int besti = 0;
  for (int s = 0; s < F.nseg; ++s) {
    const SegDev& S = F.seg[s];
    const int stride = kRec64Stride[S.type];
    for (int i = 0; i < S.count; ++i)
      resolve(F, hit_any64_from(S.type, S.rec64 + (size_t)i * stride, F.o, q, d), S.first + i, best, besti);
  }
  float rgb[3], aux[6];
  const bool want_aux = F.normal_out || F.pos_out;
  shade_pixel_t<true>(F, d, best, besti, rgb, want_aux ? aux : nullptr, nullptr, org);
  if (live) store_pixel(F, c, r, rgb, best, besti, image, depth, nearest, want_aux ? aux : nullptr);
}

// ------------------------------------------------------------------------------------------------
// k_shadow_shade: the torch backend's `shadow=True` (torch/renderer.py:291-314) as a second pass over a rendered
// frame.  Per hit pixel and light a ray from the fragment towards the light, started 0.1 along it, against EVERY
// primitive in fp64 (arbitrary origins and directions: no screen-space structure to exploit, and the reference is
// all-pairs too); the light counts as visible unless a primitive other than the fragment's own is hit before the
// light.  Then the pixel is shaded again with the visibility bits and the image is overwritten.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_shadow_shade(FrameDev F, float* __restrict__ image,
                                                       const float* __restrict__ depth,
                                                       const int32_t* __restrict__ nearest,
                                                       uint64_t* __restrict__ visibility) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int r = F.row0 + blockIdx.y * 4 + threadIdx.y;
  if (c >= F.W || r >= F.row1) return;
  const size_t row = (size_t)(r - F.row0);
  const bool hit = (double)depth[row * F.depth_stride + c] <= F.far_clip;
  uint64_t vis = ~0ull;
  if (!hit) {                                               // background: nothing to shade, every bit set
    if (visibility) visibility[row * (size_t)F.W + c] = vis;
    return;
  }
  const int win = nearest[row * F.near_stride + c];
  const int s = segment_of(F, win);
  const SegDev& S = F.seg[s];
  const double* R = S.rec64 + (size_t)(win - S.first) * kRec64Stride[S.type];
  // the primary ray and its hit, exactly as the forward pass computed them
  double d[3], q0[3] = {0, 0, 0}, org[3];
  double t;
  if (F.ortho) {
    const double xs = (F.W > 1 && c == F.W - 1) ? 1.0 : (c * F.step_x + -1.0);
    const double ys = (F.H > 1 && r == F.H - 1) ? -1.0 : (r * F.step_y + 1.0);
    const double X = xs * F.half_w, Y = ys * F.half_h;
#pragma unroll
    for (int i = 0; i < 3; ++i) { q0[i] = F.bx[i] * X + F.by[i] * Y; d[i] = -F.bz[i]; }
    t = hit_any64_from(S.type, R, F.o, q0, d);
  } else {
    pixel_ray(F, c, r, d);
    t = hit_any64(S.type, R, F.o, d, true);
  }
  double p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { org[i] = F.o[i] + q0[i]; p[i] = org[i] + t * d[i]; }

  vis = 0ull;
  for (int l = 0; l < F.nlights; ++l) {
    const float* lp = F.lpos + 4 * l;
    const double v[3] = {(double)lp[0] - p[0], (double)lp[1] - p[1], (double)lp[2] - p[2]};
    const double dist = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const double dir[3] = {v[0] / dist, v[1] / dist, v[2] / dist};
    const double q[3] = {(p[0] + 0.1 * dir[0]) - F.o[0], (p[1] + 0.1 * dir[1]) - F.o[1], (p[2] + 0.1 * dir[2]) - F.o[2]};
    double tmin = __builtin_inf();
    int blocker = -1;
    for (int sg = 0; sg < F.nseg; ++sg) {
      const SegDev& B = F.seg[sg];
      const int stride = kRec64Stride[B.type];
      for (int i = 0; i < B.count; ++i) {
        const double ts = hit_any64_from(B.type, B.rec64 + (size_t)i * stride, F.o, q, dir);
        if (ts > 0.0 && ts < dist && ts < tmin) { tmin = ts; blocker = B.first + i; }   // lowest index wins ties
      }
    }
    if (blocker < 0 || blocker == win) vis |= 1ull << l;
  }
  float rgb[3];
  shade_pixel_t<true>(F, d, t, win, rgb, nullptr, nullptr, org, vis);
  float* px = image + row * F.img_stride + 3 * (size_t)c;
  px[0] = rgb[0]; px[1] = rgb[1]; px[2] = rgb[2];
  if (visibility) visibility[row * (size_t)F.W + c] = vis;
}

// ------------------------------------------------------------------------------------------------
// k_render_fast<P>: fp32 screen-space reject per pair, fp64 confirmation of the survivors.
// A wave owns 64*P consecutive pixels of one row: lane l holds columns c0 + l + 64*j, j < P.  Reject
// records are wave-uniform reads (scalar loads); the survivor branch is entered by a wave only when one
// of its 64*P pixels passes the reject test, which for small primitives is a fraction of a percent of
// the primitives, so the loop is bound by ~3 VALU operations per pair.
// ------------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ void confirm(const FrameDev& F, const SegDev& S, int i, int r, int cbase,
                                        const float (&q)[P], bool ge_zero, double (&best)[P], int (&besti)[P]) {
  const double* R = S.rec64 + (size_t)i * kRec64Stride[S.type];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const bool cand = ge_zero ? (q[j] >= 0.0f) : (q[j] <= 0.0f);
    const int c = cbase + 64 * j;
    if (cand && c < F.W) {
      double d[3];
      pixel_ray(F, c, r, d);
      resolve(F, hit_any64(S.type, R, F.o, d, F.shading != 0), S.first + i, best[j], besti[j]);
    }
  }
}

template <int P>
__global__ __launch_bounds__(256) void k_render_fast(FrameDev F, float* __restrict__ image,
                                                      float* __restrict__ depth, int32_t* __restrict__ nearest) {
  const int cbase = blockIdx.x * (64 * P) + threadIdx.x;
  const int r_raw = F.row0 + blockIdx.y * 4 + threadIdx.y;
  const bool row_live = r_raw < F.row1;
  const int r = row_live ? r_raw : F.row1 - 1;
  const float rf = (float)r;
  float cf[P];
  double best[P];
  int besti[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    cf[j] = (float)(cbase + 64 * j);
    best[j] = __builtin_inf();
    besti[j] = 0;
  }

  for (int s = 0; s < F.nseg; ++s) {
    const SegDev& S = F.seg[s];
    if (S.type == SRH_PRIM_DISK || S.type == SRH_PRIM_SPHERE) {
      for (int i = 0; i < S.count; ++i) {
        const float* Q = S.rec32 + (size_t)i * kRec32Stride[SRH_PRIM_DISK];
        const float dr = rf - Q[1];
        float q[P];
        float m = __builtin_inff();
        {
          const float e = Q[3] * dr;
          const float g = __builtin_fmaf(Q[4] * dr, dr, -1.0f);
#pragma unroll
          for (int j = 0; j < P; ++j) {
            const float dc = cf[j] - Q[0];
            q[j] = __builtin_fmaf(dc, __builtin_fmaf(Q[2], dc, e), g);
            m = fminf(m, q[j]);
          }
        }
        if (m <= 0.0f) confirm<P>(F, S, i, r, cbase, q, false, best, besti);
      }
    } else if (S.type == SRH_PRIM_TRIANGLE) {
      for (int i = 0; i < S.count; ++i) {
        const float* Q = S.rec32 + (size_t)i * kRec32Stride[SRH_PRIM_TRIANGLE];
        const float r0 = __builtin_fmaf(Q[1], rf, Q[2]);
        const float r1 = __builtin_fmaf(Q[5], rf, Q[6]);
        const float r2 = __builtin_fmaf(Q[9], rf, Q[10]);
        float q[P];
        float m = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const float e0 = __builtin_fmaf(Q[0], cf[j], r0);
          const float e1 = __builtin_fmaf(Q[4], cf[j], r1);
          const float e2 = __builtin_fmaf(Q[8], cf[j], r2);
          q[j] = fminf(fminf(e0, e1), e2);
          m = fmaxf(m, q[j]);
        }
        if (m >= 0.0f) confirm<P>(F, S, i, r, cbase, q, true, best, besti);
      }
    } else {
      float q[P];
#pragma unroll
      for (int j = 0; j < P; ++j) q[j] = 0.0f;
      for (int i = 0; i < S.count; ++i) confirm<P>(F, S, i, r, cbase, q, false, best, besti);
    }
  }

#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int c = cbase + 64 * j;
    if (c < F.W) {      // wave-divergent only in the last column block
      double d[3];
      pixel_ray(F, c, r, d);
      float rgb[3], aux[6];
      const bool want_aux = F.normal_out || F.pos_out;
      shade_pixel(F, d, best[j], besti[j], rgb, want_aux ? aux : nullptr);
      if (row_live) store_pixel(F, c, r_raw, rgb, best[j], besti[j], image, depth, nearest, want_aux ? aux : nullptr);
    }
  }
}

template <int P>
void launch_fast(const FrameDev& F, hipStream_t st, float* image, float* depth, int32_t* nearest) {
  const dim3 block(64, 4), grid((F.W + 64 * P - 1) / (64 * P), (F.row1 - F.row0 + 3) / 4);
  hipLaunchKernelGGL(k_render_fast<P>, grid, block, 0, st, F, image, depth, nearest);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct WsLayout {
  size_t off64[SRH_MAX_SEGMENTS];
  size_t off32[SRH_MAX_SEGMENTS];
  size_t lights64, frame, tilerange, neardist, counters, large, entries, entries_words;
  size_t counters_bytes;
  int tiles_x, tiles_y_max;
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
  // tile-list offsets are 32-bit and every primitive may own up to kMaxTilesPerPrim list entries
  if (total > 0xffffffffLL / kMaxTilesPerPrim)
    return fail(SRH_E_RANGE, "too many primitives (%lld): at most %lld per scene", total, 0xffffffffLL / kMaxTilesPerPrim);
  return SRH_OK;
}

// The layout depends on the primitive counts and on the full frame size only (never on the row slab), so
// one workspace serves every slab of a frame.
WsLayout layout_for(const SrhObjects* ob, int width, int height) {
  WsLayout L;
  size_t off = 0, total = 0;
  L.lights64 = off;
  off = align_up(off + (size_t)SRH_MAX_LIGHTS * 6 * sizeof(double));
  L.frame = off;                                  // the frame's own constants, for the render kernel (FrameDev::self)
  off = align_up(off + sizeof(FrameDev));
  for (int s = 0; s < ob->n_segments; ++s) {
    const SrhSegment& g = ob->seg[s];
    L.off64[s] = off;
    off = align_up(off + (size_t)g.count * kRec64Stride[g.type] * sizeof(double));
    L.off32[s] = off;
    off = align_up(off + (size_t)g.count * kRec32Stride[g.type] * sizeof(float));
    total += (size_t)g.count;
  }
  L.tiles_x = (width + kTile - 1) / kTile;
  L.tiles_y_max = (height + kTile - 1) / kTile;
  const size_t ntiles = ((size_t)L.tiles_x * L.tiles_y_max + 3) / 4 * 4;
  L.tilerange = off;
  off = align_up(off + total * 4 * sizeof(uint16_t));
#ifdef SRH_WS_SHIFT      // measurement build: extra bytes in front of the counters / lists (the layout moves the frame time)
  off = align_up(off + (size_t)(SRH_WS_SHIFT));
#endif
  L.counters = off;
  L.counters_bytes = (kCounterPad + SRH_MAX_SEGMENTS * ntiles) * sizeof(uint32_t);
  off = align_up(off + L.counters_bytes);
  L.large = off;
  off = align_up(off + total * sizeof(uint32_t));
  L.entries = off;
  // one-pass binning: every bin owns entries_words / nbins slots (at least kMinBinCap of them)
  L.entries_words = std::max(total * kMaxTilesPerPrim, (size_t)SRH_MAX_SEGMENTS * ntiles * kMinBinCap);
  off = align_up(off + L.entries_words * sizeof(uint32_t));
  L.neardist = off;                                // light views only (FrameDev::neardist); last, so that the regions a
  off = align_up(off + total * sizeof(float));     // primary frame touches keep their places
  L.total = off;
  return L;
}

// numpy/renderer.py:145-163 + numpy/ops.py:88-115 on the host, in fp64, same operation order.
int camera_to_frame(const SrhCamera* cam, FrameDev* F, bool orthonormal = false) {
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
  for (int i = 0; i < 3; ++i) { z[i] /= zl; y[i] = cam->up_is_unit ? cam->up[i] : cam->up[i] / ul; }
  double x[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
  if (orthonormal) {
    // torch backend (torch/utils.py:402-427): x = unit(cross(unit(up), z)), y = cross(z, x)
    const double xl = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (!(xl > 0)) return fail(SRH_E_CAMERA, "degenerate camera: up is parallel to the view direction");
    for (int i = 0; i < 3; ++i) x[i] /= xl;
    y[0] = z[1] * x[2] - z[2] * x[1];
    y[1] = z[2] * x[0] - z[0] * x[2];
    y[2] = z[0] * x[1] - z[1] * x[0];
  }
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
  // pixel_ray's shared-reciprocal division is the IEEE division bit for bit as long as no operand needs rescaling:
  // true for ray components that are 0 or at least ~2^-150 in magnitude and lengths within 2^+-150, which camera
  // parameters of ordinary size guarantee (a cancelled component is 0 or >= 2^-53 of its terms)
  auto ordinary = [](double v) { const double a = std::fabs(v); return a == 0.0 || (a >= 1e-15 && a <= 1e15); };
  bool ok = ordinary(F->half_w) && F->half_w != 0.0 && ordinary(F->half_h) && F->half_h != 0.0 &&
            ordinary(F->focal) && F->focal != 0.0;
  for (int i = 0; i < 3; ++i) ok = ok && ordinary(F->bx[i]) && ordinary(F->by[i]) && ordinary(F->bz[i]);
#ifdef SRH_ABL_NODIVSHARE
  ok = false;
#endif
  F->div_shared = ok ? 1 : 0;
  F->ortho = cam->ortho ? 1 : 0;
  return SRH_OK;
}

int check_rows(const FrameDev& F, int row0, int row1) {
  if (row0 < 0 || row1 > F.H || row0 >= row1)
    return fail(SRH_E_RANGE, "row range [%d,%d) outside the %d image rows", row0, row1, F.H);
  return SRH_OK;
}

// Validation and per-frame constants shared by the forward and the backward entry points.
int setup_frame(const SrhCamera* camera, const SrhObjects* objects, const SrhLights* lights,
                const SrhMaterials* materials, const SrhParams* params, void* workspace, size_t workspace_bytes,
                FrameDev* Fp, WsLayout* Lp) {
  FrameDev& F = *Fp;
  memset(&F, 0, sizeof(F));
  if (!params) return fail(SRH_E_NULL, "params is NULL");
  if (params->shading != SRH_SHADING_NUMPY && params->shading != SRH_SHADING_TORCH)
    return fail(SRH_E_TYPE, "unknown shading model %d", params->shading);
  int rc = camera_to_frame(camera, &F, params->shading == SRH_SHADING_TORCH);
  if (rc) return rc;
  if ((rc = check_objects(objects))) return rc;
  if (!lights || !materials) return fail(SRH_E_NULL, "lights / materials is NULL");
  if ((rc = check_rows(F, params->row0, params->row1))) return rc;
  if (lights->n_lights < 0 || lights->n_lights > SRH_MAX_LIGHTS)
    return fail(SRH_E_RANGE, "n_lights = %d, expected 0..%d", lights->n_lights, SRH_MAX_LIGHTS);
  if (lights->n_lights > 0 && (!lights->pos || !lights->color_idx || !lights->colors || lights->n_colors < 1))
    return fail(SRH_E_NULL, "lights arrays missing");
  if (materials->n_materials < 1 || !materials->albedo) return fail(SRH_E_NULL, "materials.albedo missing");
  if (params->mode < SRH_MODE_AUTO || params->mode > SRH_MODE_BINNED) return fail(SRH_E_TYPE, "unknown mode %d", params->mode);
  const WsLayout L = layout_for(objects, F.W, F.H);
  *Lp = L;
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
  F.shading = params->shading;
  F.double_sided = params->double_sided ? 1 : 0;
  F.use_quartic = params->use_quartic ? 1 : 0;
  F.latt = lights->attenuation;
  F.ambient = lights->ambient;
  F.coeffs = materials->coeffs;
  F.normal_out = params->normal_out;
  F.pos_out = params->pos_out;
  F.lights64 = (const double*)((char*)workspace + L.lights64);
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
  return SRH_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int srh_abi_version(void) { return SRH_ABI_VERSION; }

const char* srh_last_error(void) { return g_err; }

size_t srh_workspace_bytes(const SrhObjects* objects, int32_t width, int32_t height) {
  if (check_objects(objects) != SRH_OK) return 0;
  if (width < 1 || height < 1 || (int64_t)((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile) > 65535LL * 65535LL ||
      (width + kTile - 1) / kTile > 65535 || (height + kTile - 1) / kTile > 65535) {
    fail(SRH_E_RANGE, "frame size %d x %d out of range", width, height);
    return 0;
  }
  return layout_for(objects, width, height).total;
}

int srh_generate_rays(const SrhCamera* camera, int32_t row0, int32_t row1, float* ray_dir, void* stream) {
  FrameDev F;
  memset(&F, 0, sizeof(F));
  int rc = camera_to_frame(camera, &F);
  if (rc == SRH_OK && F.ortho) rc = fail(SRH_E_CAMERA, "srh_generate_rays: perspective cameras only");
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

// Tile-binning fields of a frame whose primitive records live in `workspace` (layout L).
static void setup_binning(FrameDev& F, const WsLayout& L, void* workspace) {
    F.tiles_x = L.tiles_x;
  F.tiles_y = (F.row1 - F.row0 + kTile - 1) / kTile;
  F.ntiles = F.tiles_x * F.tiles_y;
  F.ntiles_pad = (F.ntiles + 3) / 4 * 4;
  F.nbins = F.nseg * F.ntiles_pad;
  F.bin_cap = (int32_t)std::min<size_t>(L.entries_words / (size_t)F.nbins, 1u << 20);
  char* ws = (char*)workspace;
  F.tilerange = (uint16_t*)(ws + L.tilerange);
  F.neardist = (float*)(ws + L.neardist);
  F.counters = (uint32_t*)(ws + L.counters);
  F.large = (uint32_t*)(ws + L.large);
  F.entries = (uint32_t*)(ws + L.entries);
  F.slab_cull = 0;
  if (F.row0 > 0 || F.row1 < F.H) {
    // rows of [D0 Dc Dr]^-1 via the adjugate (cross products)
    const PixelBasis B = pixel_basis(F);
    const double* p0 = B.D0; const double* pc = B.Dc; const double* pr = B.Dr;
    const double cx[3] = {pc[1] * pr[2] - pc[2] * pr[1], pc[2] * pr[0] - pc[0] * pr[2], pc[0] * pr[1] - pc[1] * pr[0]};
    const double cg[3] = {p0[1] * pc[2] - p0[2] * pc[1], p0[2] * pc[0] - p0[0] * pc[2], p0[0] * pc[1] - p0[1] * pc[0]};
    const double det = p0[0] * cx[0] + p0[1] * cx[1] + p0[2] * cx[2];
    if (std::isfinite(det) && std::fabs(det) > 0.0) {
      for (int k = 0; k < 3; ++k) { F.slab_ma[k] = cx[k] / det; F.slab_mg[k] = cg[k] / det; }
      F.slab_na = std::sqrt(F.slab_ma[0] * F.slab_ma[0] + F.slab_ma[1] * F.slab_ma[1] + F.slab_ma[2] * F.slab_ma[2]) * 1.000001;
      F.slab_ng = std::sqrt(F.slab_mg[0] * F.slab_mg[0] + F.slab_mg[1] * F.slab_mg[1] + F.slab_mg[2] * F.slab_mg[2]) * 1.000001;
      F.slab_cull = std::isfinite(F.slab_na) && std::isfinite(F.slab_ng) ? 1 : 0;
    }
  }
}

int srh_render_fwd(const SrhCamera* camera, const SrhObjects* objects, const SrhLights* lights,
                   const SrhMaterials* materials, const SrhParams* params, void* workspace,
                   size_t workspace_bytes, float* image, float* depth, int32_t* nearest, void* stream) {
  FrameDev F;
  WsLayout L;
  int rc = setup_frame(camera, objects, lights, materials, params, workspace, workspace_bytes, &F, &L);
  if (rc) return rc;
  if (!image || !depth) return fail(SRH_E_NULL, "image / depth is NULL");
  if (F.ortho && params->shading != SRH_SHADING_TORCH)
    return fail(SRH_E_CAMERA, "orthographic projection exists only under SRH_SHADING_TORCH");
  // orthographic frames take the all-pairs fp64 kernel of their own, whatever mode is asked for
  const int mode = F.ortho ? SRH_MODE_EXACT : (params->mode == SRH_MODE_AUTO ? SRH_MODE_BINNED : params->mode);
  hipStream_t st = (hipStream_t)stream;
  // SrhParams.stages splits the frame for callers that pipeline it over two streams: SRH_STAGE_BIN = per-frame records
  // and tile bins into the workspace, SRH_STAGE_RENDER = the render kernel from bins a previous SRH_STAGE_BIN call with
  // the same arguments left there.  0 = both.
  const int stages = params->stages == 0 ? (SRH_STAGE_BIN | SRH_STAGE_RENDER) : params->stages;
  if (stages & ~(SRH_STAGE_BIN | SRH_STAGE_RENDER | SRH_STAGE_KEEP_BINS)) return fail(SRH_E_TYPE, "unknown stages mask %d", params->stages);
  if ((stages & (SRH_STAGE_BIN | SRH_STAGE_RENDER)) != (SRH_STAGE_BIN | SRH_STAGE_RENDER) && mode != SRH_MODE_BINNED)
    return fail(SRH_E_TYPE, "SrhParams.stages splits binned frames only");
  const bool abl_skip_binning = !(stages & SRH_STAGE_BIN), abl_skip_render = !(stages & SRH_STAGE_RENDER);
  if (mode == SRH_MODE_BINNED) setup_binning(F, L, workspace);
  F.keep_bins = (stages & SRH_STAGE_KEEP_BINS) ? 1 : 0;
#if SRH_FRAME_MEM
  if (mode == SRH_MODE_BINNED) F.self = (FrameDev*)((char*)workspace + L.frame);
#endif
#ifdef SRH_ALWAYS_ZERO      // measurement build: the clearing launch of every frame, as before ABI 10
  const bool counters_clean = false;
#else
  const bool counters_clean = params->counters_clean != 0;
#endif
  if (mode == SRH_MODE_BINNED && !abl_skip_binning && !counters_clean) {
    // The bin counters start every frame at zero.  The render kernel leaves them that way (render_binned_body), so
    // a workspace that goes from frame to frame needs this launch only the first time (SrhParams.counters_clean).
    // A kernel, not hipMemsetAsync: captured into a hipGraph and replayed beside a live RCCL process group a memset
    // NODE did not take effect.
    const size_t ncount = (size_t)kCounterPad + (size_t)F.nbins;
    hipLaunchKernelGGL(k_zero_counters, dim3((unsigned)((ncount + 1023) / 1024)), dim3(256), 0, st, F.counters, (uint32_t)ncount);
  }

  for (int s = 0; s < F.nseg && !abl_skip_binning; ++s) {
    launch_prep(F, s, st);
  }
  if (mode == SRH_MODE_BINNED && !abl_skip_binning) {
#if !SRH_FUSE_BIN
    hipLaunchKernelGGL(k_bin_count, dim3((unsigned)(((size_t)F.total * kCountLanes + kBinBlock - 1) / kBinBlock)), dim3(kBinBlock), 0, st, F);
#endif
  }
  if (params->ev_start) (void)hipEventRecord((hipEvent_t)params->ev_start, st);
  if (abl_skip_render) {
  } else if (mode == SRH_MODE_BINNED) {
    const unsigned groups = binned_grid(F);   // whole regions of tiles, a multiple of 8 of them (see k_render_binned)
    // one wave per tile while that still gives every SIMD several waves; four waves per tile for small frames / slabs
    const bool split = (params->waves_per_tile == 1 || params->waves_per_tile == 4) ? params->waves_per_tile == 4
                                                                                   : binned_waves_per_tile(F) == 4;
    const dim3 g4(groups * 4), b4(256), g1(groups * (4 / kWavesPerGroup1)), b1(64 * kWavesPerGroup1);
    // one object batch of a known type: the instantiation without per-batch generality and without the other types' code
    const int batch = F.nseg == 1 ? F.seg[0].type : -1;
#if SRH_FRAME_MEM
    // the render kernel reads the frame's constants from the workspace: k_prep of batch 0 put them there, unless this
    // call renders from bins an earlier call made
    if (abl_skip_binning) hipLaunchKernelGGL(k_put_frame, dim3(1), dim3(64), 0, st, F);
    const FrameConstPtr Fc = (FrameConstPtr)F.self;
#define SRH_RENDER_KERNEL k_render_binned_mem
#define SRH_RENDER_FRAME Fc
#else
#define SRH_RENDER_KERNEL k_render_binned
#define SRH_RENDER_FRAME F
#endif
#define SRH_LAUNCH_BINNED(TCH_, WPT_, G_, B_)                                                                          \
    switch (batch) {                                                                                                   \
      case SRH_PRIM_DISK: hipLaunchKernelGGL((SRH_RENDER_KERNEL<TCH_, WPT_, SRH_PRIM_DISK>), G_, B_, 0, st, SRH_RENDER_FRAME, image, depth, nearest); break;       \
      case SRH_PRIM_PLANE: hipLaunchKernelGGL((SRH_RENDER_KERNEL<TCH_, WPT_, SRH_PRIM_PLANE>), G_, B_, 0, st, SRH_RENDER_FRAME, image, depth, nearest); break;     \
      case SRH_PRIM_SPHERE: hipLaunchKernelGGL((SRH_RENDER_KERNEL<TCH_, WPT_, SRH_PRIM_SPHERE>), G_, B_, 0, st, SRH_RENDER_FRAME, image, depth, nearest); break;   \
      case SRH_PRIM_TRIANGLE: hipLaunchKernelGGL((SRH_RENDER_KERNEL<TCH_, WPT_, SRH_PRIM_TRIANGLE>), G_, B_, 0, st, SRH_RENDER_FRAME, image, depth, nearest); break; \
      default: hipLaunchKernelGGL((SRH_RENDER_KERNEL<TCH_, WPT_, -1>), G_, B_, 0, st, SRH_RENDER_FRAME, image, depth, nearest); break;  \
    }
    if (F.shading) {
      if (split) { SRH_LAUNCH_BINNED(true, 4, g4, b4) } else { SRH_LAUNCH_BINNED(true, 1, g1, b1) }
    } else {
      if (split) { SRH_LAUNCH_BINNED(false, 4, g4, b4) } else { SRH_LAUNCH_BINNED(false, 1, g1, b1) }
    }
#undef SRH_LAUNCH_BINNED
#undef SRH_RENDER_KERNEL
#undef SRH_RENDER_FRAME
  } else if (mode == SRH_MODE_EXACT) {
    const dim3 block(64, 4), grid((F.W + 63) / 64, (F.row1 - F.row0 + 3) / 4);
    if (F.ortho) hipLaunchKernelGGL(k_render_ortho, grid, block, 0, st, F, image, depth, nearest);
    else hipLaunchKernelGGL(k_render_exact, grid, block, 0, st, F, image, depth, nearest);
  } else if (F.W >= 2048) {
    launch_fast<8>(F, st, image, depth, nearest);
  } else if (F.W >= 512) {
    launch_fast<4>(F, st, image, depth, nearest);
  } else {
    launch_fast<1>(F, st, image, depth, nearest);
  }
  if (params->ev_stop) (void)hipEventRecord((hipEvent_t)params->ev_stop, st);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SRH_OK : hip_fail(e, "render launch");
}

// ---- many views of one scene per call ---------------------------------------------------------------------------
namespace {
size_t views_header_bytes(int n_views) { return align_up((size_t)n_views * sizeof(FrameDev)); }
// Frame descriptors of the batches in flight, PER DEVICE: a ring of kViewRing slots, each a pinned host staging area, a
// range of that device's constant-memory array g_view_frames (read by the render kernel) and an event that says "the
// batch that used this slot has finished".  The other kernels read the copy at the head of the caller's workspace.
// This is the one piece of state the library owns (include/srh.h, Conventions); it is created on first use of a
// device, lives until the process ends, and one mutex per device serialises claim + submit.
constexpr int kMaxDevices = 64;
struct ViewRing {
  std::mutex mu;
  FrameDev* stage[kViewRing] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t done[kViewRing] = {nullptr, nullptr, nullptr, nullptr};
  unsigned next = 0;
};
ViewRing g_rings[kMaxDevices];

// the device a call works on: the stream's own device when HIP can tell, else the calling thread's current device
int device_of(hipStream_t st, int* dev) {
  if (st && hipStreamGetDevice(st, dev) == hipSuccess) return SRH_OK;
  const hipError_t e = hipGetDevice(dev);
  return e == hipSuccess ? SRH_OK : hip_fail(e, "hipGetDevice");
}
}  // namespace

size_t srh_workspace_bytes_views(const SrhObjects* objects, int32_t width, int32_t height, int32_t n_views) {
  const size_t one = srh_workspace_bytes(objects, width, height);
  if (!one) return 0;
  if (n_views < 1 || n_views > kMaxViewsPerCall) {
    fail(SRH_E_RANGE, "n_views = %d, expected 1..%d per call", n_views, kMaxViewsPerCall);
    return 0;
  }
  return views_header_bytes(n_views) + (size_t)n_views * one;
}

int srh_render_views(int32_t n_views, const SrhCamera* cameras, const SrhObjects* objects, const SrhLights* lights,
                     const SrhMaterials* materials, const SrhParams* params, void* workspace, size_t workspace_bytes,
                     float* images, float* depths, int32_t* nearests, void* stream) {
  if (!cameras || !params || !workspace) return fail(SRH_E_NULL, "cameras / params / workspace is NULL");
  if (!images || !depths) return fail(SRH_E_NULL, "images / depths is NULL");
  if (n_views < 1 || n_views > kMaxViewsPerCall)
    return fail(SRH_E_RANGE, "n_views = %d, expected 1..%d per call", n_views, kMaxViewsPerCall);
  if (params->mode != SRH_MODE_AUTO && params->mode != SRH_MODE_BINNED)
    return fail(SRH_E_TYPE, "srh_render_views renders in the binned mode only");
  if (params->normal_out || params->pos_out || params->ev_start || params->ev_stop)
    return fail(SRH_E_TYPE, "srh_render_views: normal / pos outputs and event hooks are per-frame features");
  const int W = cameras[0].viewport[2] - cameras[0].viewport[0], H = cameras[0].viewport[3] - cameras[0].viewport[1];
  const size_t one = srh_workspace_bytes(objects, W, H);
  if (!one) return SRH_E_RANGE;               // srh_workspace_bytes left the message
  if (params->per_view & ~(SRH_VIEWS_OBJECTS | SRH_VIEWS_LIGHTS | SRH_VIEWS_MATERIALS))
    return fail(SRH_E_TYPE, "unknown per_view mask %d", params->per_view);
  if (!lights || !materials) return fail(SRH_E_NULL, "lights / materials is NULL");
  // view v's scene: the shared structs, or element v of the arrays SrhParams.per_view names
  auto objects_of = [&](int v) { return (params->per_view & SRH_VIEWS_OBJECTS) ? objects + v : objects; };
  auto lights_of = [&](int v) { return (params->per_view & SRH_VIEWS_LIGHTS) ? lights + v : lights; };
  auto materials_of = [&](int v) { return (params->per_view & SRH_VIEWS_MATERIALS) ? materials + v : materials; };
  if (params->per_view & SRH_VIEWS_OBJECTS)
    for (int v = 1; v < n_views; ++v) {     // one workspace layout, one kernel instantiation and one grid for the whole batch
      if (objects[v].n_segments != objects[0].n_segments)
        return fail(SRH_E_RANGE, "view %d has %d object batches, view 0 has %d", v, objects[v].n_segments, objects[0].n_segments);
      for (int s = 0; s < objects[0].n_segments; ++s)
        if (objects[v].seg[s].type != objects[0].seg[s].type || objects[v].seg[s].count != objects[0].seg[s].count)
          return fail(SRH_E_RANGE, "view %d, batch %d: type %d x %d, view 0 has type %d x %d", v, s, objects[v].seg[s].type,
                      objects[v].seg[s].count, objects[0].seg[s].type, objects[0].seg[s].count);
    }
  if (params->per_view & SRH_VIEWS_LIGHTS)
    for (int v = 1; v < n_views; ++v)
      if (lights[v].n_lights != lights[0].n_lights)
        return fail(SRH_E_RANGE, "view %d has %d lights, view 0 has %d", v, lights[v].n_lights, lights[0].n_lights);
  const size_t head = views_header_bytes(n_views);
  if (workspace_bytes < head + (size_t)n_views * one)
    return fail(SRH_E_RANGE, "workspace holds %zu bytes, %d views need %zu", workspace_bytes, n_views,
                head + (size_t)n_views * one);
  hipStream_t st = (hipStream_t)stream;
  if (cameras[0].ortho) {
    // Orthographic views (torch semantics): each view is the all-pairs fp64 frame of k_render_ortho -- its frame
    // constants travel as kernel arguments, so this branch needs no staging, no ring and no lock.
    if (params->shading != SRH_SHADING_TORCH)
      return fail(SRH_E_CAMERA, "orthographic projection exists only under SRH_SHADING_TORCH");
    char* wso = (char*)workspace;
    const size_t rows = (size_t)(params->row1 - params->row0);
    for (int v = 0; v < n_views; ++v) {
      if (!cameras[v].ortho) return fail(SRH_E_CAMERA, "view %d is perspective, view 0 orthographic: one projection per call", v);
      const int w = cameras[v].viewport[2] - cameras[v].viewport[0], h = cameras[v].viewport[3] - cameras[v].viewport[1];
      if (w != W || h != H) return fail(SRH_E_RANGE, "view %d is %d x %d, view 0 is %d x %d", v, w, h, W, H);
      FrameDev F;
      WsLayout Lo;
      SrhParams pv = *params;
      if (params->view_row0) { pv.row0 = params->view_row0[v]; pv.row1 = pv.row0 + (params->row1 - params->row0); }
      int rc = setup_frame(&cameras[v], objects_of(v), lights_of(v), materials_of(v), &pv, wso + head + (size_t)v * one, one, &F, &Lo);
      if (rc) return rc;
      for (int s = 0; s < F.nseg; ++s) {
        launch_prep(F, s, st);
      }
      const dim3 block(64, 4), grid((F.W + 63) / 64, (F.row1 - F.row0 + 3) / 4);
      hipLaunchKernelGGL(k_render_ortho, grid, block, 0, st, F, images + (size_t)v * rows * F.img_stride,
                         depths + (size_t)v * rows * F.depth_stride,
                         nearests ? nearests + (size_t)v * rows * F.near_stride : nullptr);
    }
    hipError_t eo = hipGetLastError();
    return eo == hipSuccess ? SRH_OK : hip_fail(eo, "ortho views launch");
  }
  int dev = 0;
  if (int rc = device_of(st, &dev)) return rc;
  if (dev < 0 || dev >= kMaxDevices) return fail(SRH_E_RANGE, "device %d: srh_render_views supports devices 0..%d", dev, kMaxDevices - 1);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != dev)
    return fail(SRH_E_RANGE, "srh_render_views: the stream belongs to device %d but device %d is current", dev, cur);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return fail(SRH_E_TYPE, "srh_render_views cannot be stream-captured (it waits on an event and stages through "
                            "library-owned pinned memory); capture srh_render_fwd calls instead");
  ViewRing& ring = g_rings[dev];
  std::lock_guard<std::mutex> lock(ring.mu);
  // claim the next ring slot; its previous batch must have finished (host wait only when kViewRing batches are behind)
  const unsigned slot = ring.next % kViewRing;
  if (!ring.stage[slot]) {
    // staging first, then the event: a slot is usable only when it has both
    FrameDev* mem = nullptr;
    const hipError_t em = hipHostMalloc((void**)&mem, (size_t)kMaxViewsPerCall * sizeof(FrameDev), hipHostMallocDefault);
    if (em != hipSuccess) return hip_fail(em, "hipHostMalloc(frames)");
    hipEvent_t ev = nullptr;
    const hipError_t ee = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (ee != hipSuccess) { (void)hipHostFree(mem); return hip_fail(ee, "hipEventCreate"); }
    ring.stage[slot] = mem;
    ring.done[slot] = ev;
  } else {
    const hipError_t es = hipEventSynchronize(ring.done[slot]);
    if (es != hipSuccess) return hip_fail(es, "hipEventSynchronize(slot)");
  }
  FrameDev* stage = ring.stage[slot];
  char* ws = (char*)workspace;
  WsLayout L;
  for (int v = 0; v < n_views; ++v) {
    const int w = cameras[v].viewport[2] - cameras[v].viewport[0], h = cameras[v].viewport[3] - cameras[v].viewport[1];
    if (w != W || h != H) return fail(SRH_E_RANGE, "view %d is %d x %d, view 0 is %d x %d", v, w, h, W, H);
    FrameDev& F = stage[v];
    SrhParams pv = *params;
    if (params->view_row0) {                        // this view's own rows, same count for every view
      pv.row0 = params->view_row0[v];
      pv.row1 = pv.row0 + (params->row1 - params->row0);
    }
    int rc = setup_frame(&cameras[v], objects_of(v), lights_of(v), materials_of(v), &pv, ws + head + (size_t)v * one, one, &F, &L);
    if (rc) return rc;
    if (F.ortho) return fail(SRH_E_CAMERA, "view %d is orthographic, view 0 perspective: one projection per call", v);
    setup_binning(F, L, ws + head + (size_t)v * one);
  }
  const FrameDev* Fs = (const FrameDev*)ws;
  hipError_t e = hipMemcpyAsync(ws, stage, (size_t)n_views * sizeof(FrameDev), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(frames)");
  const int base = (int)slot * kMaxViewsPerCall;
  e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_view_frames), stage, (size_t)n_views * sizeof(FrameDev),
                             (size_t)base * sizeof(FrameDev), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpyToSymbolAsync(frames)");
  const FrameDev& F0 = stage[0];
  const unsigned V = (unsigned)n_views;
  const size_t ncount = (size_t)kCounterPad + (size_t)F0.nbins;
  if (!params->counters_clean)               // as in srh_render_fwd: every view's render kernel leaves its counters at zero
    hipLaunchKernelGGL(k_views_zero, dim3((unsigned)((ncount + 255) / 256), V), dim3(256), 0, st, Fs);
  for (int s = 0; s < F0.nseg; ++s)
    launch_prep_views(F0, Fs, s, V, st);
#if !SRH_FUSE_BIN
  hipLaunchKernelGGL(k_bin_count_views, dim3((unsigned)(((size_t)F0.total * kCountLanes + kBinBlock - 1) / kBinBlock), V), dim3(kBinBlock), 0, st, Fs);
#endif
  const unsigned groups = binned_grid(F0);
  // all views share the GPU, so the batch as a whole decides the launch shape
  const bool split = (params->waves_per_tile == 1 || params->waves_per_tile == 4)
                         ? params->waves_per_tile == 4 : (size_t)F0.ntiles * V < (size_t)SRH_SPLIT_TILES;
  {
    const dim3 g4(groups * 4, V), b4(256), g1(groups * (4 / kWavesPerGroup1), V), b1(64 * kWavesPerGroup1);
    const int batch = F0.nseg == 1 ? F0.seg[0].type : -1;        // as in srh_render_fwd: the typed instantiation
#define SRH_LAUNCH_VIEWS(TCH_, WPT_, G_, B_)                                                                            \
    switch (batch) {                                                                                                   \
      case SRH_PRIM_DISK: hipLaunchKernelGGL((k_render_binned_views<TCH_, WPT_, SRH_PRIM_DISK>), G_, B_, 0, st, base, images, depths, nearests); break;       \
      case SRH_PRIM_PLANE: hipLaunchKernelGGL((k_render_binned_views<TCH_, WPT_, SRH_PRIM_PLANE>), G_, B_, 0, st, base, images, depths, nearests); break;     \
      case SRH_PRIM_SPHERE: hipLaunchKernelGGL((k_render_binned_views<TCH_, WPT_, SRH_PRIM_SPHERE>), G_, B_, 0, st, base, images, depths, nearests); break;   \
      case SRH_PRIM_TRIANGLE: hipLaunchKernelGGL((k_render_binned_views<TCH_, WPT_, SRH_PRIM_TRIANGLE>), G_, B_, 0, st, base, images, depths, nearests); break; \
      default: hipLaunchKernelGGL((k_render_binned_views<TCH_, WPT_, -1>), G_, B_, 0, st, base, images, depths, nearests); break;  \
    }
    if (F0.shading) {
      if (split) { SRH_LAUNCH_VIEWS(true, 4, g4, b4) } else { SRH_LAUNCH_VIEWS(true, 1, g1, b1) }
    } else {
      if (split) { SRH_LAUNCH_VIEWS(false, 4, g4, b4) } else { SRH_LAUNCH_VIEWS(false, 1, g1, b1) }
    }
#undef SRH_LAUNCH_VIEWS
  }
  // the slot is consumed only now: a call that failed validation above leaves the ring as it was
  const hipError_t er = hipEventRecord(ring.done[slot], st);
  ring.next++;
  if (er != hipSuccess) {
    // without the event nothing says when the staging may be reused: wait here, once, rather than race later
    (void)hipStreamSynchronize(st);
    return hip_fail(er, "hipEventRecord(slot)");
  }
  e = hipGetLastError();
  return e == hipSuccess ? SRH_OK : hip_fail(e, "views launch");
}

namespace {
// Workspace of the accelerated shadow pass: the primary frame's layout first (so a workspace sized by
// srh_workspace_bytes still serves the all-pairs fallback), then the light views' frame descriptors, the scene bounds,
// and one kShadowRes^2 binning slice per light.
struct ShadowLayout {
  WsLayout primary, slice;
  size_t frames, bounds, slices, total;
};
ShadowLayout shadow_layout_for(const SrhObjects* ob, int width, int height, int n_lights) {
  ShadowLayout S;
  S.primary = layout_for(ob, width, height);
  S.slice = layout_for(ob, kShadowRes, kShadowRes);
  size_t off = S.primary.total;
  S.frames = off;
  off = align_up(off + (size_t)SRH_MAX_LIGHTS * sizeof(FrameDev));
  S.bounds = off;
  off = align_up(off + 64 * sizeof(int));
  S.slices = off;
  S.total = off + (size_t)(n_lights > 0 ? n_lights : 0) * S.slice.total;
  return S;
}
}  // namespace

size_t srh_shadow_workspace_bytes(const SrhObjects* objects, int32_t width, int32_t height, int32_t n_lights) {
  if (!srh_workspace_bytes(objects, width, height)) return 0;        // validates, leaves the message
  if (n_lights < 0 || n_lights > SRH_MAX_LIGHTS) { fail(SRH_E_RANGE, "n_lights = %d, expected 0..%d", n_lights, SRH_MAX_LIGHTS); return 0; }
  return shadow_layout_for(objects, width, height, n_lights).total;
}

int srh_shadow_shade(const SrhCamera* camera, const SrhObjects* objects, const SrhLights* lights,
                     const SrhMaterials* materials, const SrhParams* params, void* workspace, size_t workspace_bytes,
                     const int32_t* nearest, const float* depth, float* image, uint64_t* visibility, void* stream) {
  FrameDev F;
  WsLayout L;
  int rc = setup_frame(camera, objects, lights, materials, params, workspace, workspace_bytes, &F, &L);
  if (rc) return rc;
  if (!nearest || !depth || !image) return fail(SRH_E_NULL, "nearest / depth / image is NULL");
  if (params->shading != SRH_SHADING_TORCH)
    return fail(SRH_E_TYPE, "shadow rays belong to SRH_SHADING_TORCH (the numpy backend has none)");
  hipStream_t st = (hipStream_t)stream;
  // the workspace may have served other frames since the forward pass: rebuild the fp64 records (no binning)
  for (int s = 0; s < F.nseg; ++s) {
    launch_prep(F, s, st);
  }
  const dim3 block(64, 4), grid((F.W + 63) / 64, (F.row1 - F.row0 + 3) / 4);
  const ShadowLayout SL = shadow_layout_for(objects, F.W, F.H, F.nlights);
  const bool accelerated = params->mode != SRH_MODE_EXACT && F.nlights > 0 && workspace_bytes >= SL.total;
  if (!accelerated) {
    // all pairs (mode = SRH_MODE_EXACT asks for it; a workspace sized by srh_workspace_bytes has no room for the views)
    hipLaunchKernelGGL(k_shadow_shade, grid, block, 0, st, F, image, depth, nearest, visibility);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SRH_OK : hip_fail(e, "shadow launch");
  }
  char* ws = (char*)workspace;
  FrameDev* frames = (FrameDev*)(ws + SL.frames);
  int* bounds = (int*)(ws + SL.bounds);
  // template of a light view: the primary frame's inputs and shading, a kShadowRes^2 viewport, slice 0's pointers
  FrameDev T = F;
  {
    SrhCamera dummy;
    memset(&dummy, 0, sizeof(dummy));
    dummy.eye[2] = 1.0; dummy.eye[3] = 1.0; dummy.at[3] = 1.0; dummy.up[1] = 1.0;
    dummy.fovy = 1.5707963267948966; dummy.focal_length = 1.0; dummy.near_clip = 1.0e-300; dummy.far_clip = 1.0e300;
    dummy.viewport[2] = kShadowRes; dummy.viewport[3] = kShadowRes;
    if ((rc = camera_to_frame(&dummy, &T, true))) return rc;
  }
  T.row0 = 0;
  T.row1 = kShadowRes;
  T.normal_out = nullptr;
  T.pos_out = nullptr;
  char* slice0 = ws + SL.slices;
  T.lights64 = (const double*)(slice0 + SL.slice.lights64);
  for (int s = 0; s < T.nseg; ++s) {
    T.seg[s].rec64 = (const double*)(slice0 + SL.slice.off64[s]);
    T.seg[s].rec32 = (const float*)(slice0 + SL.slice.off32[s]);
  }
  setup_binning(T, SL.slice, slice0);
  T.bin_pad = 1.0;              // shadow rays cross the image plane between pixel centres
  T.near_ball = 0.1001;         // occluders up to 0.1 behind the light count (torch/renderer.py:299-306)
  const unsigned V = (unsigned)F.nlights;
  hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(64), 0, st, bounds);
  for (int s = 0; s < F.nseg; ++s)
    hipLaunchKernelGGL(k_scene_bounds, dim3((unsigned)std::min(kBoundsBlocks, (F.seg[s].count + 255) / 256)), dim3(256), 0, st, F, s, bounds);
  hipLaunchKernelGGL(k_light_frames, dim3((V + 63) / 64), dim3(64), 0, st, T, F.lpos, F.nlights, bounds, frames, SL.slice.total);
  const size_t ncount = (size_t)kCounterPad + (size_t)T.nbins;
  hipLaunchKernelGGL(k_views_zero, dim3((unsigned)((ncount + 255) / 256), V), dim3(256), 0, st, frames);
  for (int s = 0; s < T.nseg; ++s)
    launch_prep_views(T, frames, s, V, st);
#if !SRH_FUSE_BIN
  hipLaunchKernelGGL(k_bin_count_views, dim3((unsigned)(((size_t)T.total * kCountLanes + kBinBlock - 1) / kBinBlock), V), dim3(kBinBlock), 0, st, frames);
#endif
  hipLaunchKernelGGL(k_shadow_shade_binned, grid, block, 0, st, F, frames, image, depth, nearest, visibility);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SRH_OK : hip_fail(e, "shadow launch");
}

int srh_render_bwd(const SrhCamera* camera, const SrhObjects* objects, const SrhLights* lights,
                   const SrhMaterials* materials, const SrhParams* params, void* workspace, size_t workspace_bytes,
                   const float* grad_image, const float* grad_depth, const int32_t* nearest, const float* depth,
                   const SrhGrads* grads, void* stream) {
  FrameDev F;
  WsLayout L;
  int rc = setup_frame(camera, objects, lights, materials, params, workspace, workspace_bytes, &F, &L);
  if (rc) return rc;
  if (!grad_image || !nearest || !depth || !grads)
    return fail(SRH_E_NULL, "grad_image / nearest / depth / grads is NULL");
  if (F.ortho && params->shading != SRH_SHADING_TORCH)
    return fail(SRH_E_CAMERA, "orthographic projection exists only under SRH_SHADING_TORCH");
  GradsDev G;
  for (int s = 0; s < SRH_MAX_SEGMENTS; ++s) {
    G.pos[s] = grads->pos[s]; G.normal[s] = grads->normal[s]; G.radius[s] = grads->radius[s]; G.face[s] = grads->face[s];
  }
  G.lights_pos = grads->lights_pos;
  G.colors = grads->colors;
  G.albedo = grads->albedo;
  const bool tch = params->shading == SRH_SHADING_TORCH;
  G.coeffs = tch ? grads->coeffs : nullptr;
  G.attenuation = tch ? grads->attenuation : nullptr;
  G.ambient = tch ? grads->ambient : nullptr;
  hipStream_t st = (hipStream_t)stream;
  // the workspace may have served other frames since the forward pass: rebuild the fp64 records (no binning)
  for (int s = 0; s < F.nseg; ++s) {
    launch_prep(F, s, st);
  }
  const dim3 block(64, 4), grid((F.W + 63) / 64, (F.row1 - F.row0 + 3) / 4);
  if (params->ev_start) (void)hipEventRecord((hipEvent_t)params->ev_start, st);
  if (tch) hipLaunchKernelGGL(k_render_bwd_tch, grid, block, 0, st, F, G, grad_image, grad_depth, nearest, depth,
                              (const uint64_t*)params->visibility);
  else hipLaunchKernelGGL(k_render_bwd, grid, block, 0, st, F, G, grad_image, grad_depth, nearest, depth);
  if (params->ev_stop) (void)hipEventRecord((hipEvent_t)params->ev_stop, st);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SRH_OK : hip_fail(e, "backward launch");
}

int srh_bin_counters(const SrhObjects* objects, int32_t width, int32_t height, int32_t row0, int32_t row1,
                     size_t* offset_bytes, int32_t* tiles_x, int32_t* tiles_y, int32_t* ntiles_pad, int32_t* bin_cap) {
  if (!offset_bytes || !tiles_x || !tiles_y || !ntiles_pad || !bin_cap) return fail(SRH_E_NULL, "an output pointer is NULL");
  if (!srh_workspace_bytes(objects, width, height)) return SRH_E_RANGE;         // validates, leaves the message
  if (row0 < 0 || row1 > height || row0 >= row1) return fail(SRH_E_RANGE, "row range [%d,%d) outside the %d image rows", row0, row1, height);
  const WsLayout L = layout_for(objects, width, height);
  FrameDev F;
  memset(&F, 0, sizeof(F));
  F.W = width; F.H = height; F.row0 = row0; F.row1 = row1; F.nseg = objects->n_segments;
  setup_binning(F, L, nullptr);                        // pointers relative to NULL: only the tile arithmetic is used
  *offset_bytes = L.counters;
  *tiles_x = F.tiles_x; *tiles_y = F.tiles_y; *ntiles_pad = F.ntiles_pad; *bin_cap = F.bin_cap;
  return SRH_OK;
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

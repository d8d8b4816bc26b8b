// Device-side data model and the fp64 "truth" functions of the hip render backend (gfx950 only).
//
// Everything that decides a pixel's value funnels through the functions in this header:
//   pixel_ray()      -- generate_rays          (reference numpy/renderer.py:145-169)
//   hit_*64()        -- ray_*_intersection     (reference numpy/renderer.py:9-130)
//   shade_pixel()    -- fragment stage, clip, tonemap (reference numpy/renderer.py:234-263)
// They mirror the reference's fp64 formulas term by term (the TU is compiled with
// -ffp-contract=off so a*b+c is not fused behind our back); fp32 only appears when the three
// outputs are stored.  Faster render modes may *skip* calling hit_*64 for pairs they can prove are
// misses, but never replace it, which is why all modes are bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "srh.h"

namespace srh {

constexpr int kRec64Stride[4] = {8, 4, 4, 24};   // doubles per primitive record, by SRH_PRIM_*
constexpr int kTile = 16;                         // binning tile edge in pixels
constexpr int kMaxTilesPerPrim = 64;              // primitives overlapping more tiles go to the `large` list
constexpr int kMinBinCap = 16;                    // one-pass binning: smallest capacity of a bin list
constexpr int kCounterPad = 64;                   // dwords in front of the per-tile counters
// The frame-wide (`large`) list lengths exist twice, and frames alternate between the two sets: every tile of a frame
// reads its set until the render kernel ends, so that kernel cannot zero it for the next frame -- but it can zero the
// OTHER set, which nobody reads any more, and hand it over: word kLargeNext says which set the next frame's binning
// counts into (flipped by the render kernel), word kLargeNow which set this frame's render kernel reads (copied there
// by the binning).  All in device memory: a captured frame can be replayed any number of times.
constexpr int kLargeNext = 8, kLargeNow = 9;

// One scene['objects'] entry as the kernels see it.
struct SegDev {
  int32_t type, count, first, pad;   // first = global index of the segment's first primitive
  const double* rec64;               // per-frame records written by k_prep (layout below)
  const float* rec32;                // per-frame fp32 reject records (FAST mode)
  const float* pos;
  const float* normal;
  const float* radius;
  const float* face;
  const int32_t* mat;
};

// Per-frame constants, passed to kernels by value.
struct FrameDev {
  double o[3];                       // ray origin = eye[:3]
  double bx[3], by[3], bz[3];        // camera basis columns of lookat_inv (x not unit in general, Q1)
  double half_w, half_h, focal;      // w/2, h/2, focal_length
  double step_x, step_y;             // linspace steps: 2/(W-1), -2/(H-1) (0 when the axis has 1 sample)
  double near_clip, far_clip, gamma;
  int32_t W, H, row0, row1;
  int32_t nseg, total, nlights, ncolors, nmat, tonemap;
  int64_t img_stride, depth_stride, near_stride;   // elements per output row
  // SRH_SHADING_TORCH extras (torch/renderer.py:82-125)
  int32_t shading, double_sided, use_quartic;
  int32_t div_shared;                // pixel_ray may share one reciprocal between its three divisions (see there)
  int32_t ortho, pad4;               // orthographic projection (torch semantics only; k_render_ortho)
  const float* latt;                 // (L,3) attenuation or NULL
  const float* coeffs;               // (K,3) material coefficients or NULL
  const float* ambient;              // (3) or NULL
  float* normal_out;                 // optional (rows,W,3)
  float* pos_out;                    // optional (rows,W,3)
  // tile binning (BINNED mode): 16x16-pixel tiles over the rendered row slab
  int32_t tiles_x, tiles_y, ntiles, ntiles_pad;   // ntiles_pad = ntiles rounded up to a multiple of 4
  int32_t nbins, bin_cap;                         // nbins = nseg * ntiles_pad; bin = seg * ntiles_pad + tile; entries per bin list
  // Row pre-cull of a slab render (binned mode, row0 > 0 or row1 < H): x - eye = a D0 + b Dc + g Dr puts a point on
  // image row g / a; slab_ma / slab_mg are the rows of [D0 Dc Dr]^-1 that give a and g, slab_na / slab_ng their
  // lengths.  slab_cull = 0 switches the test off (full frame, or a singular basis).
  double slab_ma[3], slab_mg[3], slab_na, slab_ng;
  int32_t slab_cull, pad3;
  uint16_t* tilerange;               // (total,4) tx0,ty0,tx1,ty1 inclusive; tx0 > tx1 = not binned
  uint32_t* counters;                // [4p + s] n_large of batch s, set p | [8] set of the NEXT frame | [9] set of this frame |
                                     // [64, 64+nbins) bin counts   (see kLargeNext below)
  uint32_t* large;                   // (total) primitives too big to bin; batch s owns [seg[s].first, +count)
  uint32_t* entries;                 // (nbins, bin_cap) binned global indices: bin b's list starts at b * bin_cap
  // Light views of the shadow pass (srh_shadow.h): bins are queried at CONTINUOUS positions, so a tile's rectangle grows
  // by bin_pad pixels on every side (0 for pixel-centre rendering), and primitives within near_ball of the eye go to
  // the `large` lists (every query tests them).  view_valid = 0: this light has no usable view (all-pairs fallback).
  double bin_pad, near_ball;
  float* neardist;                   // light views: (total) lower bound of the distance from the eye (= the light) to any
                                     // point of the primitive, 0 where none is known (srh_shadow.h: the skip in front
                                     // of the exact test)
  int32_t view_valid;
  int32_t keep_bins;                 // the render kernel leaves the bin counters alone (SRH_STAGE_KEEP_BINS, light views)
  SegDev seg[SRH_MAX_SEGMENTS];
  const double* lights64;            // (nlights,6) per-frame fp64 copy written by k_prep: position xyz, colour rgb
  FrameDev* self;                    // binned frames: where in the workspace k_prep leaves a copy of this struct for the
                                     // render kernel (k_render_binned reads its constants from there, see srh_binned.h)
  const float* lpos;
  const int32_t* lcidx;
  const float* colors;
  const float* albedo;
};

// rec64 layouts
//   disk     [0..2] n^ (unit normal)  [3] k = sum(pos*n^) - n^.eye  [4..6] c = pos  [7] r^2
//   plane    [0..2] n^                [3] k
//   sphere   [0..2] oc = eye - pos    [3] c = |oc|^2 - r^2
//   triangle [0..2] n^ [3] k [4..12] v0,v1,v2 (xyz) [13..21] e01,e12,e20 [22..23] pad

__device__ __forceinline__ double dot3(const double* a, const double* b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

// numpy/renderer.py:145-169 for pixel (column c, row r) of the full W x H grid.
// Returns |D|, the length of the un-normalised direction.
__device__ __forceinline__ double pixel_ray(const FrameDev& F, int c, int r, double d[3]) {
  // np.linspace(-1, 1, W)[c], np.linspace(1, -1, H)[r]: start + i*step, last sample forced to stop
  const double xs = (F.W > 1 && c == F.W - 1) ? 1.0 : (c * F.step_x + -1.0);
  const double ys = (F.H > 1 && r == F.H - 1) ? -1.0 : (r * F.step_y + 1.0);
  const double X = xs * F.half_w, Y = ys * F.half_h, Z = -F.focal;
  double v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = (F.bx[i] * X + F.by[i] * Y) + F.bz[i] * Z;
  const double len = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  if (F.div_shared) {
    // The three IEEE divisions by |D| as hipcc expands each of them (v_rcp_f64, two Newton steps, quotient, residual
    // correction), with the reciprocal refined ONCE: for operands that need no v_div_scale rescaling -- which the
    // host checks on the camera's magnitudes (camera_to_frame) -- every step is the same arithmetic, so the results
    // are bit-identical to v[i] / len (tools/ubench_div.hip: 0 mismatches in 2.5e9 divisions on the MI355X, over
    // 400 binary orders of magnitude).  Saves two of the three quarter-rate reciprocals and 17 instructions per ray.
    const double r = __builtin_amdgcn_rcp(len);
    const double e0 = __builtin_fma(-len, r, 1.0);
    const double r1 = __builtin_fma(r, e0, r);
    const double e1 = __builtin_fma(-len, r1, 1.0);
    const double r2 = __builtin_fma(r1, e1, r1);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double q = v[i] * r2;
      const double rem = __builtin_fma(-len, q, v[i]);
      d[i] = __builtin_fma(rem, r2, q);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = v[i] / len;
  }
  return len;
}

// |D|^2 of the un-normalised direction of pixel (c, r): the part of pixel_ray the fp32 sweep needs (its 1 / |D| comes
// from one v_rsq_f32 of this, no fp64 square root or division)
__device__ __forceinline__ double pixel_len2(const FrameDev& F, int c, int r) {
  const double xs = (F.W > 1 && c == F.W - 1) ? 1.0 : (c * F.step_x + -1.0);
  const double ys = (F.H > 1 && r == F.H - 1) ? -1.0 : (r * F.step_y + 1.0);
  const double X = xs * F.half_w, Y = ys * F.half_h, Z = -F.focal;
  double v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = (F.bx[i] * X + F.by[i] * Y) + F.bz[i] * Z;
  return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
}

// ---- intersections: return the reference's ray_distance for one (ray, primitive) pair ------------
// plane: numpy/renderer.py:53-74
__device__ __forceinline__ double hit_plane64(const double* R, const double d[3]) {
  return R[3] / dot3(R, d);
}

// disk: numpy/renderer.py:77-93 -- plane hit kept where |p - c|^2 <= r^2, else inf.  In the reference's order of
// operations, p = eye + t d first and the centre subtracted from it (:5-6, :85): with a centre ~1e15 away that rounds
// differently from (eye - c) + t d, and the rim of such a disc is decided by exactly that rounding.
__device__ __forceinline__ double hit_disk64(const double* R, const double o[3], const double d[3]) {
  const double t = R[3] / dot3(R, d);
  const double qx = (o[0] + t * d[0]) - R[4], qy = (o[1] + t * d[1]) - R[5], qz = (o[2] + t * d[2]) - R[6];
  const double dist_sqr = (qx * qx + qy * qy) + qz * qz;
  return (dist_sqr <= R[7]) ? t : __builtin_inf();
}

// sphere: numpy/renderer.py:9-50 with its sentinels (Q2): bad roots -> 1.0, missed line -> 0
__device__ __forceinline__ double hit_sphere64(const double* R, const double d[3], bool* line_hits = nullptr) {
  const double a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
  const double b = 2.0 * dot3(R, d);
  const double disc = b * b - 4.0 * a * R[3];
  const bool ok = disc >= 0.0;
  if (line_hits) *line_hits = ok;
  const double root = sqrt(ok ? disc : 0.0);
  const double inv = 1.0 / (2.0 * a);
  double t1 = (-b - root) * inv, t2 = (-b + root) * inv;
  t1 = (ok && t1 >= 0.0) ? t1 : 1.0;
  t2 = (ok && t2 >= 0.0) ? t2 : 1.0;
  const double t = fmin(t1, t2);
  return ok ? t : 0.0;
}

// triangle: numpy/renderer.py:96-130 -- plane through vertex 0 with the supplied normal, then
// dot(cross(edge_i, p - v_i), n^) >= 0 for the three edges, p = eye + t*d
__device__ __forceinline__ double hit_triangle64(const double* R, const double o[3], const double d[3]) {
  const double t = R[3] / dot3(R, d);
  const double p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
  bool inside = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double* v = R + 4 + 3 * i;
    const double* e = R + 13 + 3 * i;
    const double w[3] = {p[0] - v[0], p[1] - v[1], p[2] - v[2]};
    const double cx = e[1] * w[2] - e[2] * w[1];
    const double cy = e[2] * w[0] - e[0] * w[2];
    const double cz = e[0] * w[1] - e[1] * w[0];
    inside = inside && (((cx * R[0] + cy * R[1]) + cz * R[2]) >= 0.0);
  }
  return inside ? t : __builtin_inf();
}

// sphere under the torch backend's semantics (torch/utils.py:238-279), with its data-dependent sentinels replaced
// by what they stand for: a negative root or a missed line is a miss
__device__ __forceinline__ double hit_sphere64_tch(const double* R, const double d[3]) {
  const double a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
  const double b = 2.0 * dot3(R, d);
  const double disc = b * b - 4.0 * a * R[3];
  if (!(disc >= 0.0)) return __builtin_inf();
  const double root = sqrt(disc);
  const double t1 = (-b - root) / (2.0 * a), t2 = (-b + root) / (2.0 * a);
  const double inf = __builtin_inf();
  return fmin(t1 >= 0.0 ? t1 : inf, t2 >= 0.0 ? t2 : inf);
}

__device__ __forceinline__ double hit_any64(int type, const double* R, const double o[3], const double d[3],
                                            bool tch = false) {
  switch (type) {
    case SRH_PRIM_DISK: return hit_disk64(R, o, d);
    case SRH_PRIM_PLANE: return hit_plane64(R, d);
    case SRH_PRIM_SPHERE: return tch ? hit_sphere64_tch(R, d) : hit_sphere64(R, d);
    default: return hit_triangle64(R, o, d);
  }
}

// The same intersections for a ray that starts at eye + q instead of at the eye (orthographic projection, torch
// semantics): the eye-relative records shift by q -- k' = k - n^.q, oc' = oc + q, |oc'|^2 - r^2 = c + 2 oc.q + |q|^2
// (a disc's record holds its centre: oc = eye - centre is formed here).
__device__ __forceinline__ double hit_any64_from(int type, const double* R, const double o[3], const double q[3],
                                                 const double d[3]) {
  if (type == SRH_PRIM_SPHERE) {
    const double oc[3] = {R[0] + q[0], R[1] + q[1], R[2] + q[2]};
    const double Rs[4] = {oc[0], oc[1], oc[2], (R[3] + 2.0 * dot3(R, q)) + dot3(q, q)};
    return hit_sphere64_tch(Rs, d);
  }
  const double t = (R[3] - dot3(R, q)) / dot3(R, d);
  if (type == SRH_PRIM_PLANE) return t;
  if (type == SRH_PRIM_DISK) {
    const double qx = ((o[0] - R[4]) + q[0]) + t * d[0], qy = ((o[1] - R[5]) + q[1]) + t * d[1],
                 qz = ((o[2] - R[6]) + q[2]) + t * d[2];
    return ((qx * qx + qy * qy) + qz * qz <= R[7]) ? t : __builtin_inf();
  }
  const double p[3] = {(o[0] + q[0]) + t * d[0], (o[1] + q[1]) + t * d[1], (o[2] + q[2]) + t * d[2]};
  bool inside = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double* v = R + 4 + 3 * i;
    const double* e = R + 13 + 3 * i;
    const double w[3] = {p[0] - v[0], p[1] - v[1], p[2] - v[2]};
    const double cx = e[1] * w[2] - e[2] * w[1];
    const double cy = e[2] * w[0] - e[0] * w[2];
    const double cz = e[0] * w[1] - e[1] * w[0];
    inside = inside && (((cx * R[0] + cy * R[1]) + cz * R[2]) >= 0.0);
  }
  return inside ? t : __builtin_inf();
}

// numpy/renderer.py:219-223: a hit is valid iff near <= t <= far; the running minimum only moves on
// a strictly smaller t, so the lowest global index wins ties exactly like np.argmin.
__device__ __forceinline__ void resolve(const FrameDev& F, double t, int gidx, double& best, int& besti) {
  if (F.near_clip <= t && t <= F.far_clip && t < best) { best = t; besti = gidx; }
}

// Order-independent form of the same rule, for traversals that do not visit primitives in index order:
// lexicographic minimum of (t, global index).
__device__ __forceinline__ void resolve_lex(const FrameDev& F, double t, int gidx, double& best, int& besti) {
  if (F.near_clip <= t && t <= F.far_clip && (t < best || (t == best && gidx < besti && besti != 0x7fffffff))) {
    best = t;
    besti = gidx;
  }
}

// depth as stored: the numpy backend leaves +inf where nothing is hit (numpy/renderer.py:228), the torch backend
// far + 1 (torch/renderer.py:180-183)
__device__ __forceinline__ float background_depth(const FrameDev& F, double z) {
  return (F.shading && !(z <= F.far_clip)) ? (float)(F.far_clip + 1.0) : (float)z;
}

__device__ __forceinline__ void store_aux(const FrameDev& F, size_t row, int c, const float aux[6]) {
  const size_t p = (row * (size_t)F.W + (size_t)c) * 3;
  if (F.normal_out) { F.normal_out[p] = aux[0]; F.normal_out[p + 1] = aux[1]; F.normal_out[p + 2] = aux[2]; }
  if (F.pos_out) { F.pos_out[p] = aux[3]; F.pos_out[p + 1] = aux[4]; F.pos_out[p + 2] = aux[5]; }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 1/sqrt(x) for x > 0 without the IEEE sqrt + divide expansions: v_rsq_f64 seed (about 26 bits) and one Newton step
// (about 50 bits).  Only the fragment stage uses it; its result is rounded to fp32 on output.
__device__ __forceinline__ double rsqrt_newton(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int it = 0; it < 1; ++it) {
    const double e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
  }
  return y;
}

// 1/x for x != 0 without the IEEE division expansion: v_rcp_f64 seed and one Newton step (about 50 bits).  Fragment
// stage only.
__device__ __forceinline__ double rcp_newton(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}

// base ** expo for the specular lobe (base in [0, 1] after the relu, expo a material constant), in fp32 the way
// tonemap_f32 does it: hardware log2 / exp2 while |expo * log2 base| <= 12 (error <= 1.1e-6 relative, and the lobe is
// one term of a sum), library powf otherwise -- which also supplies 0 ** 0 = 1 and the other special values.
__device__ __forceinline__ double spec_pow_f32(double base, double expo) {
  const float x = (float)base, g = (float)expo;
  const float y = g * __builtin_amdgcn_logf(x);
  const bool direct = (x >= 1e-30f && x <= 1e30f && fabsf(y) <= 12.0f) || (x == 0.0f && g > 0.0f);
  return (double)(direct ? __builtin_amdgcn_exp2f(y) : powf(x, g));
}

// image ** gamma of the tonemap (numpy/renderer.py:140-142) evaluated in fp32: x is already within half an
// fp32 ulp of the reference value.  x^g = exp2(g * log2 x) on the hardware v_log_f32 / v_exp_f32 (1 ulp each):
// the error of y = g log2 x is at most 2^-23 |y|, so for |y| <= 12 the result is off by at most
// ln2 * 12 * 2^-23 + 2^-23 = 1.1e-6 relative (parity budget 2e-6 relative + 2e-7 absolute).
__device__ __forceinline__ float tonemap_f32(const FrameDev& F, double v) {
  if (!F.tonemap) return (float)v;
  const float x = (float)v, g = (float)F.gamma;
  const float y = g * __builtin_amdgcn_logf(x);
  // direct path iff gamma > 0 and y <= 12.  Below y = -12 the relative error of exp2(y) grows (ln2 |y| 2^-23) but the
  // result is < 2.5e-4, so the ABSOLUTE error stays under 3e-9 (budget 2e-7); x = 0 and denormal x give -inf -> 0, off
  // by at most 1e-30 absolute.  NaN, +inf and y > 12 fail the compare and take the library powf (~2 ulp, all the
  // reference's special values); gamma <= 0 (never in practice) always does.
  if (g > 0.0f && y <= 12.0f) return __builtin_amdgcn_exp2f(y);
  return powf(x, g);
}

// tonemap_f32(F, 0.0) in closed form: what every all-miss or depth-masked pixel stores.  0 ** gamma is 0 for gamma > 0
// (the direct path: exp2(gamma * -inf)), 1 for gamma == 0, +inf for gamma < 0 and NaN for a NaN gamma (powf's special
// values, the same ones numpy's ** gives).  Spelled out because hipcc hoists the inlined powf(0, gamma) -- ~190 vector
// instructions on wave-uniform values -- out of the finish loop and runs it once per TILE whether or not any pixel is
// masked: 3.3 M of the render kernel's 40 M vector instructions per launch at BASELINE config 5.
__device__ __forceinline__ float tonemap_zero(const FrameDev& F) {
  if (!F.tonemap) return 0.0f;
  const float g = (float)F.gamma;
  return g > 0.0f ? 0.0f : (g == 0.0f ? 1.0f : (g < 0.0f ? __builtin_inff() : g));
}

// Fragment stage for one pixel (numpy/renderer.py:228-263): gathers the winner's normal / position /
// albedo, Lambert over all lights with no per-light clamp (Q4, Q5), depth mask, clip, tonemap.
// `z` is +inf and `win` 0 for an all-miss pixel, exactly what np.argmin hands the reference; whatever
// garbage the reference computes there is overwritten by the depth mask (:256), so masked pixels skip
// the light loop and go straight to tonemap(0).
// Arithmetic is fp64; vectors are normalised by multiplying with rsqrt_newton(|v|^2) instead of dividing
// each component by sqrt(|v|^2) (equal to ~1e-16, immaterial after the fp32 store).
// What a caller that already fetched the winner's record can hand to the fragment stage: valid when win == g.
struct ShadeHint {
  int g;              // global primitive index the hint belongs to (-1: none)
  int m;              // its material index, clamped
  double n[3];        // its unit normal (planar types; unused for spheres)
};

template <bool TCH, int BATCH = -1>
__device__ __forceinline__ void shade_pixel_t(const FrameDev& F, const double d[3], double z, int win,
                                              float rgb[3], float aux[6] = nullptr, const ShadeHint* hint = nullptr,
                                              const double* origin = nullptr, uint64_t vis = ~0ull) {
  const bool masked = (z < F.near_clip) || (z > F.far_clip);      // :256
  if (aux) {
#pragma unroll
    for (int i = 0; i < 6; ++i) aux[i] = 0.0f;
  }
  if (masked) {
    rgb[0] = rgb[1] = rgb[2] = tonemap_zero(F);
    return;
  }
  // the winner's batch: with one batch in the scene (wave-uniform test) nothing has to be selected per lane
  int seg_type = BATCH >= 0 ? BATCH : F.seg[0].type, seg_first = F.seg[0].first;
  const float* seg_pos = F.seg[0].pos;
  const double* seg_rec64 = F.seg[0].rec64;
  const int32_t* seg_mat = F.seg[0].mat;
  if (BATCH < 0 && F.nseg > 1) {
#pragma unroll
    for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
      if (i < F.nseg && win >= F.seg[i].first) {
        seg_type = F.seg[i].type; seg_first = F.seg[i].first; seg_pos = F.seg[i].pos; seg_rec64 = F.seg[i].rec64;
        seg_mat = F.seg[i].mat;
      }
  }
  const int li = win - seg_first;
  const double* org = origin ? origin : F.o;     // orthographic rays start on the image plane, not at the eye
  // From here on the arithmetic feeds only the fp32 image (2e-7 + 2e-6 |x| against the reference): multiply-adds are
  // written as explicit FMAs -- every kernel that inlines this function then rounds identically, so the render modes
  // stay bit-identical to one another -- and the reciprocal square roots are Newton-refined hardware seeds.
  const double p[3] = {__builtin_fma(z, d[0], org[0]), __builtin_fma(z, d[1], org[1]), __builtin_fma(z, d[2], org[2])};

  double n[3];
  if (seg_type == SRH_PRIM_SPHERE) {
    // (p - c) / |p - c|, zero where the ray's line misses the sphere (:45-47); p == c gives nan as there
    const float* c = seg_pos + 4 * (size_t)li;
    const double v[3] = {p[0] - (double)c[0], p[1] - (double)c[1], p[2] - (double)c[2]};
    const double len2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const double inv = (len2 > 0.0) ? rsqrt_newton(len2) : (0.0 / len2);
    bool ok;
    (void)hit_sphere64(seg_rec64 + 4 * (size_t)li, d, &ok);
    // the torch backend normalises with its eps: v / sqrt(|v|^2 + 3e-10) (torch/utils.py:131-135)
    const double inv_t = (len2 + 3.0e-10 > 0.0) ? rsqrt_newton(len2 + 3.0e-10) : 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) n[i] = TCH ? v[i] * inv_t : (ok ? v[i] * inv : 0.0);
  } else if (hint && hint->g == win) {
    n[0] = hint->n[0]; n[1] = hint->n[1]; n[2] = hint->n[2];
  } else {
    // unit normal as k_prep normalised it (ops.normalize, zero vectors stay zero, numpy/ops.py:18-26)
    const double* R = seg_rec64 + (size_t)li * kRec64Stride[seg_type];
    n[0] = R[0]; n[1] = R[1]; n[2] = R[2];
  }
  const int m = (hint && hint->g == win) ? hint->m : clampi(seg_mat[li], 0, F.nmat - 1);
  const double alb[3] = {(double)F.albedo[3 * m], (double)F.albedo[3 * m + 1], (double)F.albedo[3 * m + 2]};
  if (aux) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { aux[i] = (float)n[i]; aux[3 + i] = (float)p[i]; }
  }

  double im[3] = {0.0, 0.0, 0.0};
  if (TCH) {
    // Phong fragment shader of the torch backend (torch/renderer.py:82-125)
    const double cf[3] = {F.coeffs ? (double)F.coeffs[3 * m] : 1.0, F.coeffs ? (double)F.coeffs[3 * m + 1] : 0.0,
                          F.coeffs ? (double)F.coeffs[3 * m + 2] : 0.0};
    const double cv[3] = {F.o[0] - p[0], F.o[1] - p[1], F.o[2] - p[2]};
    const double cinv = rsqrt_newton(((cv[0] * cv[0] + cv[1] * cv[1]) + cv[2] * cv[2]) + 3.0e-10);
    const double cdir[3] = {cv[0] * cinv, cv[1] * cinv, cv[2] * cinv};
    const double cdotn = (cdir[0] * n[0] + cdir[1] * n[1]) + cdir[2] * n[2];
    const double sgn = F.double_sided ? ((cdotn > 0.0) ? 1.0 : ((cdotn < 0.0) ? -1.0 : 0.0)) : 1.0;
    for (int l = 0; l < F.nlights; ++l) {
      const float* lp = F.lpos + 4 * l;
      const double v[3] = {(double)lp[0] - p[0], (double)lp[1] - p[1], (double)lp[2] - p[2]};
      const double len2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
      // |l| and 1 / |l| from one reciprocal square root (equal to ~1e-15, immaterial after the fp32 store)
      const double inv = (len2 > 0.0) ? rsqrt_newton(len2) : 1.0;
      const double dist = (len2 > 0.0) ? len2 * inv : 0.0;
      const double lh[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
      const double kc = F.latt ? (double)F.latt[3 * l] : 1.0, kl = F.latt ? (double)F.latt[3 * l + 1] : 0.0,
                   kq = F.latt ? (double)F.latt[3 * l + 2] : 0.0;
      const double dp = F.use_quartic ? (len2 * len2) : len2;
      const double den = (kc + dist * kl) + dp * kq;
      const double afac = (fabs(den) > 0.0) ? rcp_newton(den) : 1.0;
      const double ldn = (lh[0] * n[0] + lh[1] * n[1]) + lh[2] * n[2];
      double ndotl = sgn * (afac * ldn);
      // reflect_ray(-l^, n) = 2 (l^.n) n - l^ ;  dotted with the view direction
      double rdotc = sgn * (2.0 * ldn * cdotn - ((cdir[0] * lh[0] + cdir[1] * lh[1]) + cdir[2] * lh[2]));
      ndotl = fmax(ndotl, 0.0);
      rdotc = fmax(rdotc, 0.0);
      const double spec = (cf[1] != 0.0) ? cf[1] * spec_pow_f32(rdotc, cf[2]) : 0.0;
      // light visibility (shadow rays, :116-118) multiplies light colour x albedo, not the ambient term
      const double w = (cf[0] * ndotl + spec) * (double)((vis >> l) & 1ull);
      const int ci = clampi(F.lcidx[l], 0, F.ncolors - 1);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)      // the ambient term is added once per light, as the reference does (:116-121)
        im[ch] += w * ((double)F.colors[3 * ci + ch] * alb[ch]) + (F.ambient ? (double)F.ambient[ch] : 0.0) * alb[ch];
    }
  } else {
#ifdef SRH_SHADE_F32
    // MEASUREMENT BUILD: the light loop in fp32 on two lights at a time (packed forms); the hit
    // point and the light vectors are formed in fp64 and rounded, everything after that is fp32.  Not the product path:
    // it needs the image tolerance loosened to ~2e-6 absolute (dark pixels, where the tonemap amplifies the sum's error).
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 nx = {(float)n[0], (float)n[0]}, ny = {(float)n[1], (float)n[1]}, nz = {(float)n[2], (float)n[2]};
    f2 acc[3] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
    for (int l = 0; l < F.nlights; l += 2) {
      const double* L0 = F.lights64 + 6 * l;
      const double* L1 = F.lights64 + 6 * min(l + 1, F.nlights - 1);
      const float w1 = (l + 1 < F.nlights) ? 1.0f : 0.0f;
      const f2 vx = {(float)(L0[0] - p[0]), (float)(L1[0] - p[0])}, vy = {(float)(L0[1] - p[1]), (float)(L1[1] - p[1])},
               vz = {(float)(L0[2] - p[2]), (float)(L1[2] - p[2])};
      f2 len2 = __builtin_elementwise_fma(vz, vz, __builtin_elementwise_fma(vy, vy, vx * vx));
      len2 = len2 + f2{1.0e-37f, 1.0e-37f};
      const f2 inv = {__builtin_amdgcn_rsqf(len2[0]), __builtin_amdgcn_rsqf(len2[1])};
      const f2 nd = __builtin_elementwise_fma(nz, vz, __builtin_elementwise_fma(ny, vy, nx * vx)) * inv * f2{1.0f, w1};
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) acc[ch] = __builtin_elementwise_fma(nd, f2{(float)L0[3 + ch], (float)L1[3 + ch]}, acc[ch]);
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) im[ch] = (double)(acc[ch][0] + acc[ch][1]);
    if (false)
#endif
    // sum_l (n . l^_l) colour_l, times the albedo once at the end (the reference multiplies inside the sum: equal to
    // ~1e-16, immaterial after the fp32 store); lights come as doubles from the per-frame copy
    for (int l = 0; l < F.nlights; ++l) {
      const double* L = F.lights64 + 6 * l;
      const double v[3] = {L[0] - p[0], L[1] - p[1], L[2] - p[2]};
      const double len2 = __builtin_fma(v[2], v[2], __builtin_fma(v[1], v[1], v[0] * v[0]));
      // |l| <= 0 -> 1 (Q7): the light sits exactly on the fragment and contributes n . 0 = 0.  Clamping |l|^2 at the
      // smallest normal double gives the same 0 (0 x finite) without a compare and two selects; a NaN still ends NaN.
      const double inv = rsqrt_newton(fmax(len2, 2.2250738585072014e-308));
      const double ndotl = __builtin_fma(n[2], v[2], __builtin_fma(n[1], v[1], n[0] * v[0])) * inv;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) im[ch] = __builtin_fma(ndotl, L[3 + ch], im[ch]);
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) im[ch] *= alb[ch];
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    double v = im[ch];
    if (v < 0.0) v = 0.0;                                         // :259, NaN stays NaN
    rgb[ch] = tonemap_f32(F, v);                                  // :262-263
  }
}

// runtime dispatch on the frame's shading model (kernels on the hot path instantiate shade_pixel_t directly)
__device__ __forceinline__ void shade_pixel(const FrameDev& F, const double d[3], double z, int win,
                                            float rgb[3], float aux[6] = nullptr) {
  if (F.shading) shade_pixel_t<true>(F, d, z, win, rgb, aux);
  else shade_pixel_t<false>(F, d, z, win, rgb, aux);
}

}  // namespace srh

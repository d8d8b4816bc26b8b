// Shadow pass of the torch backend's `shadow=True` (torch/renderer.py:291-314), accelerated (gfx950).
//
// Per hit pixel and light the reference casts a ray from the fragment towards the light, started 0.1 along it, against
// EVERY primitive (O(pixels x lights x primitives); k_shadow_shade in srh.hip is that all-pairs pass in fp64 and
// stays as the checker and the fallback).  All shadow rays of one light pass through the light, so seen FROM the
// light they are the rays of a pinhole camera, and the primitives a ray can meet are those whose image in that camera
// covers the ray's image point.  That is the problem the primary pass already solves with tile bins:
//
//   k_scene_bounds          bounding box of the finite primitives (discs, spheres, triangles), on the device
//   k_light_frames          per light a FrameDev of a camera at the light looking at the box's centre, its field of
//                           view the box's bounding sphere (+2 %), kShadowRes^2 virtual pixels; no usable view (light
//                           inside or too near the box, non-finite bounds) -> that light falls back to all pairs
//   k_prep / k_bin_* views  the frame pipeline's own kernels, one "view" per light: reject shapes in the light's
//                           screen space, 16 x 16-pixel tile bins.  Two changes for this use: a tile's rectangle is
//                           grown by bin_pad, because a shadow ray meets the light's image plane BETWEEN pixel centres,
//                           and primitives within near_ball of the light go to the frame-wide lists, because the
//                           reference accepts occluders up to 0.1 behind the light (ts < |L - p| from an origin 0.1 in)
//   k_shadow_shade_binned   per hit pixel and light: the ray's image point -> its tile -> the tile's candidates (and
//                           the frame-wide lists; a ray outside the view can meet only those), minus those no point of
//                           which is close enough to the light to lie between it and the fragment (FrameDev::neardist)
//                           -> the SAME fp64 test against the SAME primary-pass records as the all-pairs kernel, until
//                           one candidate other than the pixel's own primitive blocks the light in front of the
//                           pixel's own hit (lowest index on ties).  Then the pixel is shaded again with the
//                           visibility bits.
//
// A candidate list is a superset of the primitives the ray meets (the stored shapes are inflated, the bins padded), and
// every candidate goes through the exact test, so visibility bits and image equal the all-pairs pass bit for bit.
#pragma once
#include "srh_binned.h"

namespace srh {

// Virtual pixels per side of a light view.  A shadow ray is a POINT query into the view's 16 x 16-pixel tile bins, so
// the resolution only sets how fine the bins are: finer bins mean shorter lists per query (at 2048 a bin of BASELINE
// config 5 holds ~90 candidates of which a ray's line meets ~14) but more (primitive, tile) claims in the binning -- and
// a primitive whose image exceeds 64 tiles goes to the frame-wide list that EVERY query tests.  Measured, shadow pass
// alone: config 5 (100 k discs of radius 0.02) 3.91 ms at 2048, 2.87 at 4096, 3.90 at 8192; bunny.obj at 512^2
// 0.60 / 1.31 / 18.9 ms.  So every light picks its own resolution, 2048 or 4096, from the mean size of the scene's
// primitives (k_scene_bounds): the finer one while a mean primitive stays within ~3 tiles across.
#ifndef SRH_SHADOW_RES
#define SRH_SHADOW_RES 4096
#endif
constexpr int kShadowRes = SRH_SHADOW_RES;        // the largest view: workspace slices are sized for it
constexpr int kShadowResCoarse = 2048;
constexpr double kShadowFineMaxTiles = 3.2;       // mean primitive diameter, in tiles, up to which the fine view is taken

// monotonic int encoding of a float, for atomicMin / atomicMax
__device__ __forceinline__ int ordered_int(float f) {
  const int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ordered_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// bounds: [0..2] min xyz, [3..5] max xyz (ordered ints), [6] = 1 if a non-finite extent was seen,
// [7] sum of the finite primitives' radii (float bits; a triangle counts half its longest edge), [8] their number
__global__ void k_bounds_init(int* bounds) {
  if (threadIdx.x < 3) bounds[threadIdx.x] = ordered_int(3.0e38f);
  else if (threadIdx.x < 6) bounds[threadIdx.x] = ordered_int(-3.0e38f);
  else if (threadIdx.x < 9) bounds[threadIdx.x] = 0;
}

// Grid-stride over the batch, one set of atomics per WORKGROUP (waves combine through LDS): with one set per wave of a
// thread-per-primitive grid the 100 k discs of BASELINE config 5 queued 12 k atomics on eight addresses -- 146 us.
constexpr int kBoundsBlocks = 64;
__global__ __launch_bounds__(256) void k_scene_bounds(FrameDev F, int s, int* bounds) {
  const SegDev& S = F.seg[s];
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  bool any_bad = false;
  float size = 0.0f, count = 0.0f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S.count && S.type != SRH_PRIM_PLANE; i += gridDim.x * blockDim.x) {
    float plo[3], phi[3], psize;
    bool bad = false;
    if (S.type == SRH_PRIM_TRIANGLE) {
      const float* f = S.face + 12 * (size_t)i;
      for (int k = 0; k < 3; ++k) { plo[k] = 3.0e38f; phi[k] = -3.0e38f; }
      for (int v = 0; v < 3; ++v)
        for (int k = 0; k < 3; ++k) { plo[k] = fminf(plo[k], f[4 * v + k]); phi[k] = fmaxf(phi[k], f[4 * v + k]); bad |= !isfinite(f[4 * v + k]); }
      psize = 0.5f * fmaxf(fmaxf(phi[0] - plo[0], phi[1] - plo[1]), phi[2] - plo[2]);
    } else {
      const float* c = S.pos + 4 * (size_t)i;
      const float r = fabsf(S.radius[i]);
      for (int k = 0; k < 3; ++k) { plo[k] = c[k] - r; phi[k] = c[k] + r; bad |= !isfinite(plo[k]) || !isfinite(phi[k]); }
      psize = r;
    }
    if (bad) { any_bad = true; continue; }
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], plo[k]); hi[k] = fmaxf(hi[k], phi[k]); }
    if (isfinite(psize)) { size += psize; count += 1.0f; }
  }
  // wave, then workgroup
  __shared__ float red[4][8];
  __shared__ int red_bad[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    size += __shfl_xor(size, m);
    count += __shfl_xor(count, m);
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], m)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], m)); }
  }
  const bool wave_bad = __builtin_amdgcn_ballot_w64(any_bad) != 0ull;
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) { red[wave][k] = lo[k]; red[wave][3 + k] = hi[k]; }
    red[wave][6] = size; red[wave][7] = count;
    red_bad[wave] = wave_bad ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool bad = false;
    for (int w = 1; w < 4; ++w) {
      for (int k = 0; k < 3; ++k) { red[0][k] = fminf(red[0][k], red[w][k]); red[0][3 + k] = fmaxf(red[0][3 + k], red[w][3 + k]); }
      red[0][6] += red[w][6]; red[0][7] += red[w][7];
    }
    for (int w = 0; w < 4; ++w) bad |= red_bad[w] != 0;
    if (bad) bounds[6] = 1;
    if (red[0][7] > 0.0f && isfinite(red[0][6])) {
      atomicAdd(reinterpret_cast<float*>(bounds + 7), red[0][6]);
      atomicAdd(bounds + 8, (int)red[0][7]);
    }
    for (int k = 0; k < 3; ++k) {
      if (red[0][k] < 3.0e38f) atomicMin(&bounds[k], ordered_int(red[0][k]));
      if (red[0][3 + k] > -3.0e38f) atomicMax(&bounds[3 + k], ordered_int(red[0][3 + k]));
    }
  }
}

// One thread per light: the light's camera into Fs[l].  T is the host-built template for light 0 (viewport kShadowRes^2,
// binning fields, workspace pointers of light 0's slice); light l's slice lies l * slice_bytes further on.
__global__ void k_light_frames(FrameDev T, const float* __restrict__ lpos, int nlights, const int* __restrict__ bounds,
                               FrameDev* __restrict__ Fs, size_t slice_bytes) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlights) return;
  FrameDev F = T;
  const size_t by = (size_t)l * slice_bytes;
  F.tilerange = (uint16_t*)((char*)F.tilerange + by);
  F.neardist = (float*)((char*)F.neardist + by);
  F.counters = (uint32_t*)((char*)F.counters + by);
  F.large = (uint32_t*)((char*)F.large + by);
  F.entries = (uint32_t*)((char*)F.entries + by);
  F.lights64 = (const double*)((const char*)F.lights64 + by);
  for (int s = 0; s < SRH_MAX_SEGMENTS; ++s) {
    F.seg[s].rec64 = (const double*)((const char*)F.seg[s].rec64 + (size_t)l * slice_bytes);
    F.seg[s].rec32 = (const float*)((const char*)F.seg[s].rec32 + (size_t)l * slice_bytes);
  }
  const double L[3] = {(double)lpos[4 * l], (double)lpos[4 * l + 1], (double)lpos[4 * l + 2]};
  double lo[3], hi[3];
  for (int k = 0; k < 3; ++k) { lo[k] = ordered_float(bounds[k]); hi[k] = ordered_float(bounds[3 + k]); }
  const double C[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
  const double e[3] = {hi[0] - C[0], hi[1] - C[1], hi[2] - C[2]};
  const double rad = sqrt(dot3(e, e)) * 1.01 + 1.0e-6;
  const double v[3] = {L[0] - C[0], L[1] - C[1], L[2] - C[2]};           // z axis of the camera: from `at` to `eye`
  const double dist = sqrt(dot3(v, v));
  bool ok = bounds[6] == 0 && lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2] && isfinite(rad) && isfinite(dist) &&
            isfinite(L[0] + L[1] + L[2]) && dist > 1.3 * rad + 0.2;       // half angle below ~50 degrees, light outside the box
  double z[3] = {0.0, 0.0, 1.0}, half = 1.0;
  if (ok) {
    for (int k = 0; k < 3; ++k) z[k] = v[k] / dist;
    half = tan(1.02 * asin(rad / dist));
  }
  // x = unit(cross(up, z)) with `up` the world axis least aligned with z, y = cross(z, x)  (orthonormal, as torch/utils.py:402-427)
  const int ax = (fabs(z[0]) <= fabs(z[1]) && fabs(z[0]) <= fabs(z[2])) ? 0 : (fabs(z[1]) <= fabs(z[2]) ? 1 : 2);
  double up[3] = {0.0, 0.0, 0.0};
  up[ax] = 1.0;
  double x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
  const double xl = sqrt(dot3(x, x));
  for (int k = 0; k < 3; ++k) x[k] /= xl;
  const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
  for (int k = 0; k < 3; ++k) {
    F.o[k] = ok ? L[k] : 0.0;
    F.bx[k] = x[k]; F.by[k] = y[k]; F.bz[k] = z[k];
  }
  F.focal = 1.0;
  F.half_w = F.half_h = half;
  // resolution: the template is the fine view; a light whose view would make the mean primitive span more than
  // kShadowFineMaxTiles tiles takes the coarse one (same workspace slice, fewer and larger bins)
  if (ok && kShadowRes > kShadowResCoarse) {
    const double mean = bounds[8] > 0 ? (double)__int_as_float(bounds[7]) / (double)bounds[8] : 0.0;
    const double tiles_fine = (2.0 * mean) / (2.0 * half * dist) * (double)kShadowRes / (double)kTile;
    if (!(tiles_fine <= kShadowFineMaxTiles)) {
      const long long words = (long long)F.bin_cap * (long long)F.nbins;       // this slice's bin-list words
      F.W = F.H = kShadowResCoarse;
      F.row0 = 0; F.row1 = kShadowResCoarse;
      F.step_x = 2.0 / (kShadowResCoarse - 1);
      F.step_y = -2.0 / (kShadowResCoarse - 1);
      F.tiles_x = F.tiles_y = kShadowResCoarse / kTile;
      F.ntiles = F.tiles_x * F.tiles_y;
      F.ntiles_pad = (F.ntiles + 3) / 4 * 4;
      F.nbins = F.nseg * F.ntiles_pad;
      F.bin_cap = (int32_t)min(words / (long long)F.nbins, 1ll << 20);
    }
  }
  F.near_clip = 1.0e-300;                                   // "> 0": occluders behind the light camera are not binned
  F.far_clip = 1.0e300;
  F.div_shared = 0;
  F.ortho = 0;
  F.view_valid = ok ? 1 : 0;
  Fs[l] = F;
}

// Image point (continuous pixel coordinates) of the unit direction w seen from light view LF; false if behind it.
// `len` = length of the view's un-normalised ray direction through that point (D = (focal / Zn) w).
__device__ __forceinline__ bool light_image_point(const FrameDev& LF, const double w[3], double& c, double& r, double& len) {
  const double X = dot3(w, LF.bx), Y = dot3(w, LF.by), Zn = -dot3(w, LF.bz);
  if (!(Zn > 0.0)) return false;
  const double xs = (LF.focal * X / Zn) / LF.half_w, ys = (LF.focal * Y / Zn) / LF.half_h;
  c = (xs + 1.0) / LF.step_x;
  r = (ys - 1.0) / LF.step_y;
  len = LF.focal / Zn;
  return true;
}

__global__ __launch_bounds__(256) void k_shadow_shade_binned(FrameDev F, const FrameDev* __restrict__ LFs,
                                                              float* __restrict__ image, const float* __restrict__ depth,
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
  const int sw = segment_of(F, win);
  const SegDev& SW = F.seg[sw];
  const double* RW = SW.rec64 + (size_t)(win - SW.first) * kRec64Stride[SW.type];
  // the primary ray and its hit, exactly as the forward pass computed them
  double d[3], q0[3] = {0, 0, 0}, org[3];
  double t;
  if (F.ortho) {
    const double xs = (F.W > 1 && c == F.W - 1) ? 1.0 : (c * F.step_x + -1.0);
    const double ys = (F.H > 1 && r == F.H - 1) ? -1.0 : (r * F.step_y + 1.0);
    const double X = xs * F.half_w, Y = ys * F.half_h;
#pragma unroll
    for (int i = 0; i < 3; ++i) { q0[i] = F.bx[i] * X + F.by[i] * Y; d[i] = -F.bz[i]; }
    t = hit_any64_from(SW.type, RW, F.o, q0, d);
  } else {
    pixel_ray(F, c, r, d);
    t = hit_any64(SW.type, RW, F.o, d, true);
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
    auto hit = [&](int sg, int g) {
      const SegDev& B = F.seg[sg];
      return hit_any64_from(B.type, B.rec64 + (size_t)(g - B.first) * kRec64Stride[B.type], F.o, q, dir);
    };
    // The reference lights the pixel iff the nearest blocker (lowest index on ties) is the pixel's own primitive, or
    // there is none (torch/renderer.py:306-314).  So the own primitive is tested first, and the light is blocked iff
    // some OTHER primitive's hit comes lexicographically before (ts_self, win) -- the walk stops at the first such
    // candidate instead of looking for the nearest one.
    double ts_self = __builtin_inf();
    {
      const double ts = hit(sw, win);
      if (ts > 0.0 && ts < dist) ts_self = ts;
    }
    bool blocked = false;
    auto test = [&](int sg, int g) {
      if (g == win) return;
      const double ts = hit(sg, g);
      if (ts > 0.0 && ts < dist && (ts < ts_self || (ts == ts_self && g < win))) blocked = true;
    };
    const FrameDev& LF = LFs[l];
    if (!LF.view_valid || !isfinite(dist)) {
      // no usable light view: every primitive, as the all-pairs pass does
      for (int sg = 0; sg < F.nseg; ++sg)
        for (int i = 0; i < F.seg[sg].count && !blocked; ++i) test(sg, F.seg[sg].first + i);
    } else {
      // where the ray leaves the light: direction from the light towards the fragment
      const double w[3] = {-dir[0], -dir[1], -dir[2]};
      double lc = 0.0, lr = 0.0, llen = 1.0;
      const bool front = light_image_point(LF, w, lc, lr, llen);
      const bool inside = front && lc >= 0.0 && lr >= 0.0 && lc <= (double)(LF.W - 1) && lr <= (double)(LF.H - 1);
      const int tile = inside ? ((int)lr / kTile) * LF.tiles_x + (int)lc / kTile : 0;
      // A valid blocker's hit point lies between the ray's origin (0.1 in front of the fragment) and 0.1 beyond the
      // light: closer to the light than max(dist - 0.1, 0.1).  A candidate no point of which is that close
      // (FrameDev::neardist, a lower bound rounded down; `reach` rounded up) cannot block: it is skipped before its
      // 64-byte record is fetched.  Primitives within 0.1 of the light never come here (frame-wide lists).
      const float reach = (float)fmax(dist - 0.1, 0.1) * 1.000001f;
      const float* __restrict__ nd = LF.neardist;
      for (int sg = 0; sg < F.nseg; ++sg) {
        const uint32_t* big = LF.large + LF.seg[sg].first;
        const uint32_t nbig = large_length(LF, sg);
        for (uint32_t i = 0; i < nbig && !blocked; ++i) test(sg, (int)big[i]);
        if (inside && !blocked) {
          const int bin = sg * LF.ntiles_pad + tile;
          const uint32_t* list = bin_list(LF, bin);
          uint32_t nlist = bin_length(LF, bin);
          // Two phases per round, so that the lanes of a wave (each on its own list) run the expensive exact test
          // together: every lane first skips ahead to its next surviving candidates -- four list entries and their four
          // distances per step, so that a step costs two memory round trips, not eight -- then all lanes that found one
          // test it.
          uint32_t base = 0, mask = 0;
          int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
          for (;;) {
            while (mask == 0u && base < nlist) {
              const uint32_t last = nlist - 1;
              c0 = (int)list[base];
              c1 = (int)list[min(base + 1, last)];
              c2 = (int)list[min(base + 2, last)];
              c3 = (int)list[min(base + 3, last)];
              const float n0 = nd[c0], n1 = nd[c1], n2 = nd[c2], n3 = nd[c3];
              mask = (n0 < reach ? 1u : 0u) | ((base + 1 < nlist && n1 < reach) ? 2u : 0u) |
                     ((base + 2 < nlist && n2 < reach) ? 4u : 0u) | ((base + 3 < nlist && n3 < reach) ? 8u : 0u);
              base += 4;
            }
            if (mask == 0u) break;
            const int k = __builtin_ctz(mask);
            mask &= mask - 1u;
            test(sg, k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3);
            if (blocked) break;
          }
        }
      }
    }
    if (!blocked) vis |= 1ull << l;
  }
  float rgb[3];
  shade_pixel_t<true>(F, d, t, win, rgb, nullptr, nullptr, org, vis);
  float* px = image + row * F.img_stride + 3 * (size_t)c;
  px[0] = rgb[0]; px[1] = rgb[1]; px[2] = rgb[2];
  if (visibility) visibility[row * (size_t)F.W + c] = vis;
}

}  // namespace srh

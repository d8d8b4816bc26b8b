// Analytic backward of the render path (gfx950): one thread per pixel, fp64 chain rule, fp32 atomic accumulation.
//
// What is differentiated is exactly what the forward computes for a hit pixel with winner m
// (reference numpy/renderer.py:204-263; gradient semantics = autograd through the reference's torch backend,
// torch/renderer.py:136-355 + torch/utils.py:238-366 -- selection and masks are piecewise constant):
//     planar m:  n^ = n/|n|, t = n^.(q - o) / (n^.d), q = pos | face[0]          p = o + t d
//     sphere m:  t = (-b -/+ sqrt(b^2 - 4a(|oc|^2 - r^2))) / 2a, n^ = (p - c)/|p - c|
//     shading:   im_c = sum_i (n^ . l^_i) C_ic A_c,  l^_i = (L_i - p)/|L_i - p|;  clip at 0;  out_c = im_c ^ gamma
// Accumulation: every sum is reduced over the wave before it goes out as fp32 atomics -- light / colour gradients over
// all 64 lanes (every pixel contributes to the same few addresses), albedo gradients too when the whole wave shades
// with one material, per-primitive gradients over the runs of adjacent pixels won by the same primitive (RunReduce).
#pragma once
#include "srh_device.h"

namespace srh {

struct GradsDev {
  float* pos[SRH_MAX_SEGMENTS];
  float* normal[SRH_MAX_SEGMENTS];
  float* radius[SRH_MAX_SEGMENTS];
  float* face[SRH_MAX_SEGMENTS];
  float* lights_pos;
  float* colors;
  float* albedo;
  float* coeffs;        // torch shading only
  float* attenuation;   // torch shading only
  float* ambient;       // torch shading only
};

// gamma * x^(gamma - 1) for x > 0, the tonemap's derivative, the way tonemap_f32 evaluates the tonemap itself: hardware
// log2 / exp2 in fp32 (1.1e-6 relative while |(gamma - 1) log2 x| <= 12; the gradients leave as fp32 sums and are checked
// to 2e-4 of their largest entry), the fp64 library pow outside that range.  The library call is ~250 fp64
// instructions per channel: a third of the numpy-shading backward kernel.
__device__ __forceinline__ double tonemap_slope(double x, double gamma) {
  const float xf = (float)x, e = (float)(gamma - 1.0);
  const float y = e * __builtin_amdgcn_logf(xf);
  if (xf >= 1.0e-30f && xf <= 1.0e30f && fabsf(y) <= 12.0f) return gamma * (double)__builtin_amdgcn_exp2f(y);
  return gamma * pow(x, gamma - 1.0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__device__ __forceinline__ void add3(float* dst, const double g[3]) {
  if (g[0] != 0.0) atomicAdd(dst + 0, (float)g[0]);
  if (g[1] != 0.0) atomicAdd(dst + 1, (float)g[1]);
  if (g[2] != 0.0) atomicAdd(dst + 2, (float)g[2]);
}

__device__ __forceinline__ void add3(float* dst, const float g[3]) {
  if (g[0] != 0.0f) atomicAdd(dst + 0, g[0]);
  if (g[1] != 0.0f) atomicAdd(dst + 1, g[1]);
  if (g[2] != 0.0f) atomicAdd(dst + 2, g[2]);
}

// Per-primitive gradients, reduced by key over the wave before they go out as atomics.  A wave is 64 consecutive
// pixels of one image row, so pixels won by the same primitive form RUNS of adjacent lanes: a segmented inclusive scan
// over the runs (six shuffle steps) leaves each run's sum in its last lane, and only that lane issues the atomics --
// one set per (wave, run) instead of one per pixel.  A plane that fills the frame sends 6 atomics per wave to its six
// addresses instead of 384; a primitive that re-appears after a gap in the same row simply forms
// two runs.  All 64 lanes must call this (lanes without a hit pass key -1 and zeros).
struct RunReduce {
  int lane, start;      // this lane, first lane of its run
  bool tail;            // last lane of its run: holds the run's sums after reduce()
  __device__ __forceinline__ RunReduce(int key, int lane_) : lane(lane_) {
    const int prev = __shfl_up(key, 1);
    const bool head = lane == 0 || prev != key;
    const unsigned long long heads = __builtin_amdgcn_ballot_w64(head);
    const unsigned long long upto = heads & (~0ull >> (63 - lane));          // heads at lanes <= this one (lane 0 is one)
    start = 63 - __builtin_clzll(upto);
    tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
  }
  __device__ __forceinline__ float reduce(float v) const {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float t = __shfl_up(v, off);
      if (lane - off >= start) v += t;
    }
    return v;
  }
};

// Run-reduce one pixel's primitive gradients over the wave and add the run sums: gA -> pos (disc / plane / sphere) or
// face vertex 0 (triangle), gB -> normal (planar types), g_r -> radius (sphere).  key = winner's global index, -1 = none.
__device__ __forceinline__ void scatter_primitive_grads(const GradsDev& G, int key, int lane, int s, int type, int li,
                                                        const double gA[3], const double gB[3], double g_r) {
  const RunReduce run(key, lane);
  float v[7];
#pragma unroll
  for (int k = 0; k < 3; ++k) { v[k] = run.reduce((float)gA[k]); v[3 + k] = run.reduce((float)gB[k]); }
  v[6] = run.reduce((float)g_r);
  // Second level, for primitives larger than a wave's 64 pixels: when each of the workgroup's four waves (four image
  // rows) is ONE run of the same primitive, wave 0 adds the four sums and issues the only atomics of the workgroup.
  // (All 256 threads reach this barrier: a workgroup leaves the backward kernels only as a whole, at their top.)
  __shared__ float wsum[4][8];
  __shared__ int wkey[4];
  const int wave = threadIdx.y;
  const bool whole = run.start == 0 && lane == 63;          // this lane closes a run that spans the whole wave
  if (lane == 63) {
    wkey[wave] = (run.start == 0) ? key : -2;
#pragma unroll
    for (int k = 0; k < 7; ++k) wsum[wave][k] = v[k];
  }
  __syncthreads();
  const bool merged = wkey[0] >= 0 && wkey[0] == wkey[1] && wkey[0] == wkey[2] && wkey[0] == wkey[3];
  if (merged) {
    if (!(whole && wave == 0)) return;
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = (wsum[0][k] + wsum[1][k]) + (wsum[2][k] + wsum[3][k]);
  } else if (!run.tail || key < 0) {
    return;
  }
  float* dstA = nullptr;
  float* dstB = nullptr;
  float* dstR = nullptr;
#pragma unroll
  for (int i = 0; i < SRH_MAX_SEGMENTS; ++i)
    if (s == i) {
      dstA = type == SRH_PRIM_TRIANGLE ? (G.face[i] ? G.face[i] + 12 * (size_t)li : nullptr)
                                       : (G.pos[i] ? G.pos[i] + 4 * (size_t)li : nullptr);
      dstB = (type != SRH_PRIM_SPHERE && G.normal[i]) ? G.normal[i] + 4 * (size_t)li : nullptr;
      dstR = (type == SRH_PRIM_SPHERE && G.radius[i]) ? G.radius[i] + li : nullptr;
    }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (dstA && v[k] != 0.0f) atomicAdd(dstA + k, v[k]);
    if (dstB && v[3 + k] != 0.0f) atomicAdd(dstB + k, v[3 + k]);
  }
  if (dstR && v[6] != 0.0f) atomicAdd(dstR, v[6]);
}

__global__ __launch_bounds__(256) void k_render_bwd(FrameDev F, GradsDev G, const float* __restrict__ grad_image,
                                                     const float* __restrict__ grad_depth,
                                                     const int32_t* __restrict__ nearest,
                                                     const float* __restrict__ depth) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int r = F.row0 + blockIdx.y * 4 + threadIdx.y;
  const int lane = threadIdx.x;
  const bool live = (c < F.W) && (r < F.row1);
  const size_t row = live ? (size_t)(r - F.row0) : 0;
  const int cc = live ? c : 0;
  // background pixels (depth = +inf) and pixels outside the slab take part in the wave reductions with zeros
  const bool hit = live && isfinite(depth[row * F.depth_stride + cc]);
  // a workgroup (4 rows x 64 pixels) without a single hit pixel contributes nothing: it leaves together, before the
  // reductions below (whose barriers every thread of a workgroup that stays still reaches) -- a mesh that covers a fifth
  // of the frame then runs a fifth of the workgroups through the fp64 chain rule
  if (!__syncthreads_or(hit ? 1 : 0)) return;

  double g_out[3] = {0, 0, 0}, g_dep = 0.0;
  int win = 0;
  if (hit) {
    const float* gi = grad_image + row * F.img_stride + 3 * (size_t)cc;
    g_out[0] = gi[0]; g_out[1] = gi[1]; g_out[2] = gi[2];
    if (grad_depth) g_dep = grad_depth[row * F.depth_stride + cc];
    win = nearest[row * F.near_stride + cc];
  }
  int s = 0;
#pragma unroll
  for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
    if (i < F.nseg && win >= F.seg[i].first) s = i;
  int type = F.seg[0].type, first = F.seg[0].first;
  const double* rec_base = F.seg[0].rec64;
  const float* pos_base = F.seg[0].pos;
  const float* nrm_base = F.seg[0].normal;
  const float* rad_base = F.seg[0].radius;
  const float* face_base = F.seg[0].face;
  const int32_t* mat_base = F.seg[0].mat;
#pragma unroll
  for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
    if (s == i) {
      type = F.seg[i].type; first = F.seg[i].first; rec_base = F.seg[i].rec64; pos_base = F.seg[i].pos;
      nrm_base = F.seg[i].normal; rad_base = F.seg[i].radius; face_base = F.seg[i].face; mat_base = F.seg[i].mat;
    }
  const int li = win - first;

  double d[3] = {0, 0, -1};
  pixel_ray(F, live ? c : 0, live ? r : F.row0, d);

  // ---- forward quantities of this pixel ---------------------------------------------------------------------
  double t = 0.0, n[3] = {0, 0, 0}, p[3] = {0, 0, 0};
  double sph_v[3] = {0, 0, 0}, sph_inv = 0.0;          // sphere: p - c and 1/|p - c|
  bool sph_ok = false;
  const double* R = rec_base + (size_t)li * kRec64Stride[type];
  if (hit) {
    t = hit_any64(type, R, F.o, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = F.o[k] + t * d[k];
    if (type == SRH_PRIM_SPHERE) {
      const float* cp = pos_base + 4 * (size_t)li;
#pragma unroll
      for (int k = 0; k < 3; ++k) sph_v[k] = p[k] - (double)cp[k];
      const double len2 = (sph_v[0] * sph_v[0] + sph_v[1] * sph_v[1]) + sph_v[2] * sph_v[2];
      (void)hit_sphere64(R, d, &sph_ok);
      sph_inv = (len2 > 0.0 && sph_ok) ? 1.0 / sqrt(len2) : 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) n[k] = sph_v[k] * sph_inv;
    } else {
      n[0] = R[0]; n[1] = R[1]; n[2] = R[2];
    }
  }
  const int m = hit ? clampi(mat_base[li], 0, F.nmat - 1) : 0;
  double alb[3] = {0, 0, 0};
  if (hit) { alb[0] = F.albedo[3 * m]; alb[1] = F.albedo[3 * m + 1]; alb[2] = F.albedo[3 * m + 2]; }

  // first pass over the lights: the image value, needed for the clip / tonemap derivative
  double im[3] = {0, 0, 0};
  for (int l = 0; l < F.nlights; ++l) {
    const float* lp = F.lpos + 4 * l;
    const double v[3] = {(double)lp[0] - p[0], (double)lp[1] - p[1], (double)lp[2] - p[2]};
    const double len2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const double inv = (len2 > 0.0) ? rsqrt_newton(len2) : 1.0;     // ~1e-15 relative: far inside the fp32 sums
    const double sdot = ((n[0] * v[0] + n[1] * v[1]) + n[2] * v[2]) * inv;
    const int ci = clampi(F.lcidx[l], 0, F.ncolors - 1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) im[ch] += (sdot * (double)F.colors[3 * ci + ch]) * alb[ch];
  }
  // d out / d im: clip passes im >= 0; gamma: gamma * im^(gamma-1) for im > 0
  double g_im[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    double w = 0.0;
    if (hit) {
      if (F.tonemap) w = (im[ch] > 0.0) ? tonemap_slope(im[ch], F.gamma) : 0.0;
      else w = (im[ch] >= 0.0) ? 1.0 : 0.0;
    }
    g_im[ch] = g_out[ch] * w;
  }

  // second pass: gradients through the shading
  double g_n[3] = {0, 0, 0}, g_p[3] = {0, 0, 0}, g_alb[3] = {0, 0, 0};
  for (int l = 0; l < F.nlights; ++l) {
    const float* lp = F.lpos + 4 * l;
    const double v[3] = {(double)lp[0] - p[0], (double)lp[1] - p[1], (double)lp[2] - p[2]};
    const double len2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const bool nz = len2 > 0.0;
    const double inv = nz ? rsqrt_newton(len2) : 1.0;
    const double lh[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
    const double sdot = (n[0] * lh[0] + n[1] * lh[1]) + n[2] * lh[2];
    const int ci = clampi(F.lcidx[l], 0, F.ncolors - 1);
    double g_s = 0.0, g_col[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const double col = (double)F.colors[3 * ci + ch];
      g_s += g_im[ch] * col * alb[ch];
      g_alb[ch] += g_im[ch] * sdot * col;
      g_col[ch] = g_im[ch] * sdot * alb[ch];
    }
    // l^ = v/|v| (|v| = 0 -> divided by 1: l^ = v, Q7): g_v = (g_lh - l^ (l^.g_lh)) / |v|, or g_lh itself
    const double g_lh[3] = {g_s * n[0], g_s * n[1], g_s * n[2]};
    const double proj = (lh[0] * g_lh[0] + lh[1] * g_lh[1]) + lh[2] * g_lh[2];
    double g_v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g_v[k] = nz ? (g_lh[k] - lh[k] * proj) * inv : g_lh[k];
      g_n[k] += g_s * lh[k];
      g_p[k] -= g_v[k];
    }
    // light position and light colour: every pixel of the wave adds to the same address -> reduce first
    if (G.lights_pos) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float sum = wave_sum((float)g_v[k]);
        if (lane == 0 && sum != 0.0f) atomicAdd(G.lights_pos + 4 * l + k, sum);
      }
    }
    if (G.colors) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float sum = wave_sum((float)g_col[ch]);
        if (lane == 0 && sum != 0.0f) atomicAdd(G.colors + 3 * ci + ch, sum);
      }
    }
  }
  if (G.albedo) {
    // one material for the whole wave (the common case) -> one atomic per channel; otherwise per lane
    const int m0 = __builtin_amdgcn_readfirstlane(m);
    const bool uniform = __builtin_amdgcn_ballot_w64(hit && m != m0) == 0ull;
    if (uniform) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float sum = wave_sum((float)g_alb[ch]);
        if (lane == 0 && sum != 0.0f) atomicAdd(G.albedo + 3 * m0 + ch, sum);
      }
    } else if (hit) {
      add3(G.albedo + 3 * m, g_alb);
    }
  }
  // ---- geometry: every lane takes part (run reduction below); lanes without a hit carry zeros -------------------------
  double gA[3] = {0, 0, 0};        // d/d pos (disc, plane, sphere) or d/d face vertex 0 (triangle)
  double gB[3] = {0, 0, 0};        // d/d normal (planar types)
  double g_r = 0.0;                // d/d radius (sphere)
  if (hit && type == SRH_PRIM_SPHERE) {
    // n^ = v/|v| with v = p - c (zero gradient where the line misses the sphere: n^ is the constant 0 there)
    const double proj = (n[0] * g_n[0] + n[1] * g_n[1]) + n[2] * g_n[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double g_v = (g_n[k] - n[k] * proj) * sph_inv;
      g_p[k] += g_v;
      gA[k] = -g_v;
    }
    const double g_t = ((g_p[0] * d[0] + g_p[1] * d[1]) + g_p[2] * d[2]) + g_dep;
    // t = (-b -/+ root)/(2a) unless it is one of the reference's constants (1.0 for a bad root, 0 for a miss)
    const double a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    const double b = 2.0 * dot3(R, d);
    const double disc = b * b - 4.0 * a * R[3];
    if (disc > 0.0) {
      const double root = sqrt(disc), inv2a = 1.0 / (2.0 * a);
      const double t1 = (-b - root) * inv2a, t2 = (-b + root) * inv2a;
      const double t1v = (t1 >= 0.0) ? t1 : 1.0, t2v = (t2 >= 0.0) ? t2 : 1.0;
      const bool use1 = t1v <= t2v;
      const bool constant = use1 ? !(t1 >= 0.0) : !(t2 >= 0.0);
      if (!constant) {
        const double sgn = use1 ? -1.0 : 1.0;
        double g_b = -g_t * inv2a;
        const double g_disc = sgn * g_t * inv2a / (2.0 * root);
        g_b += 2.0 * b * g_disc;
        const double g_cc = -4.0 * a * g_disc;
        g_r = -2.0 * (double)rad_base[li] * g_cc;
#pragma unroll
        for (int k = 0; k < 3; ++k) gA[k] -= 2.0 * R[k] * g_cc + 2.0 * d[k] * g_b;   // oc = o - c
      }
    }
  } else if (hit) {
    // planar: t = k/den, k = n^.(q - o), den = n^.d
    const double g_t = ((g_p[0] * d[0] + g_p[1] * d[1]) + g_p[2] * d[2]) + g_dep;
    const double den = dot3(R, d);
    const double g_k = g_t / den, g_den = -g_t * t / den;
    const float* qp = (type == SRH_PRIM_TRIANGLE) ? face_base + 12 * (size_t)li : pos_base + 4 * (size_t)li;
    double g_nh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      gA[k] = g_k * n[k];
      g_nh[k] = g_n[k] + g_k * ((double)qp[k] - F.o[k]) + g_den * d[k];
    }
    // n^ = nin/|nin| (4-D norm with w = 0; a zero normal stays zero and gets the gradient divided by 1)
    const float* np_ = nrm_base + 4 * (size_t)li;
    const double nin2 = (((double)np_[0] * np_[0] + (double)np_[1] * np_[1]) + (double)np_[2] * np_[2]) + (double)np_[3] * np_[3];
    const double ninv = (nin2 > 0.0) ? 1.0 / sqrt(nin2) : 1.0;
    const double proj = (nin2 > 0.0) ? (n[0] * g_nh[0] + n[1] * g_nh[1]) + n[2] * g_nh[2] : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) gB[k] = (g_nh[k] - n[k] * proj) * ninv;
  }
  scatter_primitive_grads(G, hit ? win : -1, lane, s, type, li, gA, gB, g_r);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the torch backend's semantics (SRH_SHADING_TORCH; reference torch/renderer.py:82-125,136-355):
//     n^ = n / sqrt(|n|^2 + 3e-10)  (sphere: (p - c) likewise),   c^ = (o - p) / sqrt(|o - p|^2 + 3e-10)
//     per light i:  l = L_i - p, dist = |l|, l^ = l / dist,  afac = 1 / (kc + kl dist + kq dist^pw)   (pw = 2 | 4)
//                   ndotl = relu(sgn afac (l^ . n^)),  rdotc = relu(sgn (2 (l^ . n^)(c^ . n^) - c^ . l^))
//                   im_c += (c0 ndotl + c1 rdotc^c2) C_ic A_c + amb_c A_c
//     out_c = relu(im_c) ^ gamma
// sgn = sign(c^ . n^) with double_sided, else 1 -- a constant, like the relus (threshold_backward selects, it does not
// multiply) and the nearest-hit selection.  torch.pow: 0^0 = 1, d/d exponent = 0 at base 0, d/d base = 0 at exponent 0.
// Misses and the far + 1 background carry no gradient.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef SRH_BWD_TCH_WAVES
#define SRH_BWD_TCH_WAVES 3
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SRH_BWD_TCH_WAVES))) void k_render_bwd_tch(FrameDev F, GradsDev G, const float* __restrict__ grad_image,
                                                         const float* __restrict__ grad_depth,
                                                         const int32_t* __restrict__ nearest,
                                                         const float* __restrict__ depth,
                                                         const uint64_t* __restrict__ visibility) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int r = F.row0 + blockIdx.y * 4 + threadIdx.y;
  const int lane = threadIdx.x;
  const bool live = (c < F.W) && (r < F.row1);
  const size_t row = live ? (size_t)(r - F.row0) : 0;
  const int cc = live ? c : 0;
  const bool hit = live && ((double)depth[row * F.depth_stride + cc] <= F.far_clip);
  // a workgroup (4 rows x 64 pixels) without a single hit pixel contributes nothing: it leaves together, before the
  // reductions below (whose barriers every thread of a workgroup that stays still reaches) -- a mesh that covers a fifth
  // of the frame then runs a fifth of the workgroups through the fp64 chain rule
  if (!__syncthreads_or(hit ? 1 : 0)) return;

  float g_out[3] = {0, 0, 0}, g_dep = 0.0f;               // upstream gradients are fp32
  int win = 0;
  if (hit) {
    const float* gi = grad_image + row * F.img_stride + 3 * (size_t)cc;
    g_out[0] = gi[0]; g_out[1] = gi[1]; g_out[2] = gi[2];
    if (grad_depth) g_dep = grad_depth[row * F.depth_stride + cc];
    win = nearest[row * F.near_stride + cc];
  }
  int s = 0;
#pragma unroll
  for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
    if (i < F.nseg && win >= F.seg[i].first) s = i;
  // only what the whole kernel needs stays in registers (type, local index, record); the other per-segment pointers are
  // selected again where they are used -- six 64-bit pointers held across both light loops cost 12 registers
  int type = F.seg[0].type, first = F.seg[0].first;
  const double* rec_base = F.seg[0].rec64;
#pragma unroll
  for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
    if (s == i) { type = F.seg[i].type; first = F.seg[i].first; rec_base = F.seg[i].rec64; }
  const int li = win - first;
  auto seg_ptr = [&](auto field) {
    auto ptr = field(F.seg[0]);
#pragma unroll
    for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
      if (s == i) ptr = field(F.seg[i]);
    return ptr;
  };
  // light visibility of the forward pass (shadow rays): a constant 0 / 1 factor on each light's colour x albedo term
  const uint64_t vis = (visibility && hit) ? visibility[row * (size_t)F.W + cc] : ~0ull;

  // the pixel's ray: from the eye (perspective), or from its own origin eye + q0 on the image plane with the one
  // direction -z of the camera basis (orthographic, torch/utils.py:461-468); org is what the reference calls ray_orig
  double d[3] = {0, 0, -1}, q0[3] = {0, 0, 0};
  if (F.ortho) {
    const int pc = live ? c : 0, pr = live ? r : F.row0;
    const double xs = (F.W > 1 && pc == F.W - 1) ? 1.0 : (pc * F.step_x + -1.0);
    const double ys = (F.H > 1 && pr == F.H - 1) ? -1.0 : (pr * F.step_y + 1.0);
    const double X = xs * F.half_w, Y = ys * F.half_h;
#pragma unroll
    for (int i = 0; i < 3; ++i) { q0[i] = F.bx[i] * X + F.by[i] * Y; d[i] = -F.bz[i]; }
  } else {
    pixel_ray(F, live ? c : 0, live ? r : F.row0, d);
  }
  const double org[3] = {F.o[0] + q0[0], F.o[1] + q0[1], F.o[2] + q0[2]};

  // ---- forward quantities of this pixel ---------------------------------------------------------------------
  double t = 0.0, n[3] = {0, 0, 0}, p[3] = {0, 0, 0};
  double sph_inv = 0.0;                                    // sphere: 1 / sqrt(|p - c|^2 + 3e-10)
  const double* R = rec_base + (size_t)li * kRec64Stride[type];
  if (hit) {
    t = F.ortho ? hit_any64_from(type, R, F.o, q0, d) : hit_any64(type, R, F.o, d, true);
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = org[k] + t * d[k];
    if (type == SRH_PRIM_SPHERE) {
      const float* cp = seg_ptr([](const SegDev& S) { return S.pos; }) + 4 * (size_t)li;
      double v[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] = p[k] - (double)cp[k];
      sph_inv = 1.0 / sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + 3.0e-10);
#pragma unroll
      for (int k = 0; k < 3; ++k) n[k] = v[k] * sph_inv;
    } else {
      n[0] = R[0]; n[1] = R[1]; n[2] = R[2];
    }
  }
  const int m = hit ? clampi(seg_ptr([](const SegDev& S) { return S.mat; })[li], 0, F.nmat - 1) : 0;
  float alb[3] = {0, 0, 0}, cf[3] = {1.0f, 0.0f, 0.0f}, amb[3] = {0, 0, 0};    // inputs are fp32: kept as such, widened at use
  if (hit) {
    alb[0] = F.albedo[3 * m]; alb[1] = F.albedo[3 * m + 1]; alb[2] = F.albedo[3 * m + 2];
    if (F.coeffs) { cf[0] = F.coeffs[3 * m]; cf[1] = F.coeffs[3 * m + 1]; cf[2] = F.coeffs[3 * m + 2]; }
    if (F.ambient) { amb[0] = F.ambient[0]; amb[1] = F.ambient[1]; amb[2] = F.ambient[2]; }
  }
  // view direction
  const double u[3] = {F.o[0] - p[0], F.o[1] - p[1], F.o[2] - p[2]};
  const double sc_inv = 1.0 / sqrt(((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]) + 3.0e-10);
  const double cdir[3] = {u[0] * sc_inv, u[1] * sc_inv, u[2] * sc_inv};
  const double cdotn = (cdir[0] * n[0] + cdir[1] * n[1]) + cdir[2] * n[2];
  const double sgn = F.double_sided ? ((cdotn > 0.0) ? 1.0 : ((cdotn < 0.0) ? -1.0 : 0.0)) : 1.0;
  const int pw = F.use_quartic ? 4 : 2;

  // per-light forward terms, evaluated twice (image first: its value gates the clip / tonemap derivative).
  // The light loops run in fp32: the hit point, the normal and the view direction come out of the fp64 geometry above and
  // are rounded once; everything a light contributes is then a few dozen fp32 operations whose sums leave as fp32
  // atomics anyway.  With these terms in fp64 the kernel held 168 registers (three waves per SIMD, 9 of them spilled).
  const float nf[3] = {(float)n[0], (float)n[1], (float)n[2]};
  const float cdf[3] = {(float)cdir[0], (float)cdir[1], (float)cdir[2]};
  const float cdotnf = (float)cdotn, sgnf = (float)sgn;
  struct LightTerms {
    float lh[3], dist, afac, ldn, nd, rd, cl;
    bool nz, den_ok;
  };
  auto light_terms = [&](int l, LightTerms& T) {
    const float* lp = F.lpos + 4 * l;
    // the difference is formed in fp64 (light and fragment may be far from the origin and close to each other)
    const float v[3] = {(float)((double)lp[0] - p[0]), (float)((double)lp[1] - p[1]), (float)((double)lp[2] - p[2])};
    T.dist = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    T.nz = T.dist > 0.0f;
    const float inv = T.nz ? 1.0f / T.dist : 1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) T.lh[k] = v[k] * inv;
    const float kc = F.latt ? F.latt[3 * l] : 1.0f, kl = F.latt ? F.latt[3 * l + 1] : 0.0f, kq = F.latt ? F.latt[3 * l + 2] : 0.0f;
    const float dp = (pw == 4) ? (T.dist * T.dist) * (T.dist * T.dist) : T.dist * T.dist;
    const float den = (kc + T.dist * kl) + dp * kq;
    T.den_ok = fabsf(den) > 0.0f;
    T.afac = T.den_ok ? 1.0f / den : 1.0f;
    T.ldn = (T.lh[0] * nf[0] + T.lh[1] * nf[1]) + T.lh[2] * nf[2];
    T.cl = (cdf[0] * T.lh[0] + cdf[1] * T.lh[1]) + cdf[2] * T.lh[2];
    T.nd = sgnf * (T.afac * T.ldn);
    T.rd = sgnf * (2.0f * T.ldn * cdotnf - T.cl);
  };
  // the forward pass evaluates the lobe in fp32 too (spec_pow_f32)
  auto spec_pow = [&](float rdotc) { return (rdotc == 0.0f && cf[2] == 0.0f) ? 1.0f : powf(rdotc, cf[2]); };

  float im[3] = {0, 0, 0};
  for (int l = 0; l < F.nlights; ++l) {
    LightTerms T;
    light_terms(l, T);
    const float w = (cf[0] * fmaxf(T.nd, 0.0f) + cf[1] * spec_pow(fmaxf(T.rd, 0.0f))) * (float)((vis >> l) & 1ull);
    const int ci = clampi(F.lcidx[l], 0, F.ncolors - 1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) im[ch] += w * (F.colors[3 * ci + ch] * alb[ch]) + amb[ch] * alb[ch];
  }
  float g_im[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float w = 0.0f;
    if (hit && im[ch] > 0.0f) w = F.tonemap ? (float)F.gamma * powf(im[ch], (float)(F.gamma - 1.0)) : 1.0f;
    g_im[ch] = g_out[ch] * w;
  }

  float g_nf[3] = {0, 0, 0}, g_pf[3] = {0, 0, 0};
  float g_alb[3] = {0, 0, 0}, g_cf[3] = {0, 0, 0}, g_amb[3] = {0, 0, 0};     // leave as fp32 atomics: summed in fp32
  float g_cdf[3] = {0, 0, 0}, g_cdotnf = 0.0f;
  for (int l = 0; l < F.nlights; ++l) {
    LightTerms T;
    light_terms(l, T);
    const float ndotl = fmaxf(T.nd, 0.0f), rdotc = fmaxf(T.rd, 0.0f);
    const float P = spec_pow(rdotc);
    const float vl = (float)((vis >> l) & 1ull);
    const float w = (cf[0] * ndotl + cf[1] * P) * vl;          // the visible light's weight
    const int ci = clampi(F.lcidx[l], 0, F.ncolors - 1);
    float g_w = 0.0f, g_col[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float col = F.colors[3 * ci + ch];
      g_w += g_im[ch] * col * alb[ch];
      g_alb[ch] += g_im[ch] * (w * col + amb[ch]);
      g_col[ch] = g_im[ch] * w * alb[ch];
      g_amb[ch] += g_im[ch] * alb[ch];
    }
    g_w *= vl;                                                  // d im / d (unshadowed weight)
    g_cf[0] += g_w * ndotl;
    g_cf[1] += g_w * P;
    if (rdotc > 0.0f) g_cf[2] += (g_w * cf[1] * P) * logf(rdotc);
    const float g_nd = (T.nd > 0.0f) ? g_w * cf[0] : 0.0f;
    const float g_rd = (T.rd > 0.0f && cf[2] != 0.0f) ? g_w * cf[1] * cf[2] * powf(rdotc, cf[2] - 1.0f) : 0.0f;
    const float g_ldn = g_rd * sgnf * 2.0f * cdotnf + g_nd * sgnf * T.afac;
    g_cdotnf += g_rd * sgnf * 2.0f * T.ldn;
    const float g_cl = -g_rd * sgnf;
    const float g_afac = g_nd * sgnf * T.ldn;
    const float g_den = T.den_ok ? -g_afac * T.afac * T.afac : 0.0f;
    const float kl = F.latt ? F.latt[3 * l + 1] : 0.0f, kq = F.latt ? F.latt[3 * l + 2] : 0.0f;
    const float d2 = T.dist * T.dist;
    const float dp = (pw == 4) ? d2 * d2 : d2;
    const float ddp = (pw == 4) ? 4.0f * d2 * T.dist : 2.0f * T.dist;
    const float g_dist = g_den * (kl + kq * ddp);
    float g_lh[3], g_v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g_lh[k] = g_ldn * nf[k] + g_cl * cdf[k];
      g_nf[k] += g_ldn * T.lh[k];
      g_cdf[k] += g_cl * T.lh[k];
    }
    const float proj = (T.lh[0] * g_lh[0] + T.lh[1] * g_lh[1]) + T.lh[2] * g_lh[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g_v[k] = T.nz ? (g_lh[k] - T.lh[k] * proj) / T.dist + g_dist * T.lh[k] : g_lh[k];
      g_pf[k] -= g_v[k];
    }
    if (G.lights_pos) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float sum = wave_sum(g_v[k]);
        if (lane == 0 && sum != 0.0f) atomicAdd(G.lights_pos + 4 * l + k, sum);
      }
    }
    if (G.colors) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float sum = wave_sum(g_col[ch]);
        if (lane == 0 && sum != 0.0f) atomicAdd(G.colors + 3 * ci + ch, sum);
      }
    }
    if (G.attenuation) {
      const float ga[3] = {g_den, g_den * T.dist, g_den * dp};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float sum = wave_sum(ga[k]);
        if (lane == 0 && sum != 0.0f) atomicAdd(G.attenuation + 3 * l + k, sum);
      }
    }
  }
  // back to fp64 for the geometry chain below
  double g_n[3] = {(double)g_nf[0], (double)g_nf[1], (double)g_nf[2]}, g_p[3] = {(double)g_pf[0], (double)g_pf[1], (double)g_pf[2]};
  double g_cdir[3] = {(double)g_cdf[0], (double)g_cdf[1], (double)g_cdf[2]};
  const double g_cdotn = (double)g_cdotnf;
  if (G.ambient) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float sum = wave_sum((float)g_amb[ch]);
      if (lane == 0 && sum != 0.0f) atomicAdd(G.ambient + ch, sum);
    }
  }
  {
    // one material for the whole wave (the common case) -> one atomic per component; otherwise per lane
    const int m0 = __builtin_amdgcn_readfirstlane(m);
    const bool uniform = __builtin_amdgcn_ballot_w64(hit && m != m0) == 0ull;
    if (G.albedo) {
      if (uniform) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float sum = wave_sum((float)g_alb[ch]);
          if (lane == 0 && sum != 0.0f) atomicAdd(G.albedo + 3 * m0 + ch, sum);
        }
      } else if (hit) {
        add3(G.albedo + 3 * m, g_alb);
      }
    }
    if (G.coeffs) {
      if (uniform) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float sum = wave_sum((float)g_cf[k]);
          if (lane == 0 && sum != 0.0f) atomicAdd(G.coeffs + 3 * m0 + k, sum);
        }
      } else if (hit) {
        add3(G.coeffs + 3 * m, g_cf);
      }
    }
  }
  // ---- view direction and geometry: every lane takes part (run reduction below); lanes without a hit carry zeros ------
  double gA[3] = {0, 0, 0}, gB[3] = {0, 0, 0}, g_r = 0.0;
  if (hit) {
    // c^ . n^ and c^ = u / sqrt(|u|^2 + eps), u = o - p
#pragma unroll
    for (int k = 0; k < 3; ++k) { g_cdir[k] += g_cdotn * n[k]; g_n[k] += g_cdotn * cdir[k]; }
    const double projc = (cdir[0] * g_cdir[0] + cdir[1] * g_cdir[1]) + cdir[2] * g_cdir[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) g_p[k] -= (g_cdir[k] - cdir[k] * projc) * sc_inv;
  }
  if (hit && type == SRH_PRIM_SPHERE) {
    const double proj = (n[0] * g_n[0] + n[1] * g_n[1]) + n[2] * g_n[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double g_v = (g_n[k] - n[k] * proj) * sph_inv;
      g_p[k] += g_v;
      gA[k] = -g_v;
    }
    const double g_t = ((g_p[0] * d[0] + g_p[1] * d[1]) + g_p[2] * d[2]) + g_dep;
    // oc = ray origin - centre: the record's eye-relative oc shifted by q0 (zero for perspective rays)
    const double oc[3] = {R[0] + q0[0], R[1] + q0[1], R[2] + q0[2]};
    const double cq = (R[3] + 2.0 * dot3(R, q0)) + dot3(q0, q0);
    const double a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    const double b = 2.0 * dot3(oc, d);
    const double disc = b * b - 4.0 * a * cq;
    if (disc > 0.0) {
      const double root = sqrt(disc), inv2a = 1.0 / (2.0 * a);
      const double t1 = (-b - root) * inv2a;
      const double sg = (t1 >= 0.0) ? -1.0 : 1.0;       // the smaller non-negative root
      double g_b = -g_t * inv2a;
      const double g_disc = sg * g_t * inv2a / (2.0 * root);
      g_b += 2.0 * b * g_disc;
      const double g_cc = -4.0 * a * g_disc;
      g_r = -2.0 * (double)seg_ptr([](const SegDev& S) { return S.radius; })[li] * g_cc;
#pragma unroll
      for (int k = 0; k < 3; ++k) gA[k] -= 2.0 * oc[k] * g_cc + 2.0 * d[k] * g_b;  // oc = origin - c
    }
  } else if (hit) {
    // planar: t = k/den, k = n^.(q - origin), den = n^.d
    const double g_t = ((g_p[0] * d[0] + g_p[1] * d[1]) + g_p[2] * d[2]) + g_dep;
    const double den = dot3(R, d);
    const double g_k = g_t / den, g_den = -g_t * t / den;
    const float* qp = (type == SRH_PRIM_TRIANGLE) ? seg_ptr([](const SegDev& S) { return S.face; }) + 12 * (size_t)li
                                                  : seg_ptr([](const SegDev& S) { return S.pos; }) + 4 * (size_t)li;
    double g_nh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      gA[k] = g_k * n[k];
      g_nh[k] = g_n[k] + g_k * ((double)qp[k] - org[k]) + g_den * d[k];
    }
    // n^ = nin / sqrt(|nin|^2 + 3e-10) over xyz
    const float* np_ = seg_ptr([](const SegDev& S) { return S.normal; }) + 4 * (size_t)li;
    const double nin2 = (((double)np_[0] * np_[0] + 1e-10) + ((double)np_[1] * np_[1] + 1e-10)) + ((double)np_[2] * np_[2] + 1e-10);
    const double ninv = 1.0 / sqrt(nin2);
    const double proj = (n[0] * g_nh[0] + n[1] * g_nh[1]) + n[2] * g_nh[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) gB[k] = (g_nh[k] - n[k] * proj) * ninv;
  }
  scatter_primitive_grads(G, hit ? win : -1, lane, s, type, li, gA, gB, g_r);
}

}  // namespace srh

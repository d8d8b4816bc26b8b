// Analytic backward of the render path (gfx950): one thread per pixel, fp64 chain rule, fp32 atomic accumulation.
//
// What is differentiated is exactly what the forward computes for a hit pixel with winner m
// (reference numpy/renderer.py:204-263; gradient semantics = autograd through the reference's torch backend,
// torch/renderer.py:136-355 + torch/utils.py:238-366 -- selection and masks are piecewise constant):
//     planar m:  n^ = n/|n|, t = n^.(q - o) / (n^.d), q = pos | face[0]          p = o + t d
//     sphere m:  t = (-b -/+ sqrt(b^2 - 4a(|oc|^2 - r^2))) / 2a, n^ = (p - c)/|p - c|
//     shading:   im_c = sum_i (n^ . l^_i) C_ic A_c,  l^_i = (L_i - p)/|L_i - p|;  clip at 0;  out_c = im_c ^ gamma
// Accumulation: per-primitive gradients go out as one atomic per lane and component; light / colour gradients are
// first summed over the wave with cross-lane shuffles (every pixel contributes to the same few addresses), and so
// are albedo gradients when the whole wave shades with one material.
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
};

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
    const double inv = (len2 > 0.0) ? 1.0 / sqrt(len2) : 1.0;
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
      if (F.tonemap) w = (im[ch] > 0.0) ? F.gamma * pow(im[ch], F.gamma - 1.0) : 0.0;
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
    const double inv = nz ? 1.0 / sqrt(len2) : 1.0;
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
  if (!hit) return;                                     // no shuffles below this line

  // ---- geometry ------------------------------------------------------------------------------------------------
  if (type == SRH_PRIM_SPHERE) {
    // n^ = v/|v| with v = p - c (zero gradient where the line misses the sphere: n^ is the constant 0 there)
    const double proj = (n[0] * g_n[0] + n[1] * g_n[1]) + n[2] * g_n[2];
    double g_c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double g_v = (g_n[k] - n[k] * proj) * sph_inv;
      g_p[k] += g_v;
      g_c[k] = -g_v;
    }
    const double g_t = ((g_p[0] * d[0] + g_p[1] * d[1]) + g_p[2] * d[2]) + g_dep;
    // t = (-b -/+ root)/(2a) unless it is one of the reference's constants (1.0 for a bad root, 0 for a miss)
    const double a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    const double b = 2.0 * dot3(R, d);
    const double disc = b * b - 4.0 * a * R[3];
    double g_r = 0.0;
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
        const float* rp = rad_base + li;
        g_r = -2.0 * (double)rp[0] * g_cc;
#pragma unroll
        for (int k = 0; k < 3; ++k) g_c[k] -= 2.0 * R[k] * g_cc + 2.0 * d[k] * g_b;   // oc = o - c
      }
    }
    if (G.pos[s]) add3(G.pos[s] + 4 * (size_t)li, g_c);
    if (G.radius[s] && g_r != 0.0) atomicAdd(G.radius[s] + li, (float)g_r);
    return;
  }
  // planar: t = k/den, k = n^.(q - o), den = n^.d
  const double g_t = ((g_p[0] * d[0] + g_p[1] * d[1]) + g_p[2] * d[2]) + g_dep;
  const double den = dot3(R, d);
  const double g_k = g_t / den, g_den = -g_t * t / den;
  const float* qp = (type == SRH_PRIM_TRIANGLE) ? face_base + 12 * (size_t)li : pos_base + 4 * (size_t)li;
  double g_q[3], g_nh[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    g_q[k] = g_k * n[k];
    g_nh[k] = g_n[k] + g_k * ((double)qp[k] - F.o[k]) + g_den * d[k];
  }
  // n^ = nin/|nin| (4-D norm with w = 0; a zero normal stays zero and gets the gradient divided by 1)
  const float* np_ = nrm_base + 4 * (size_t)li;
  const double nin2 = (((double)np_[0] * np_[0] + (double)np_[1] * np_[1]) + (double)np_[2] * np_[2]) + (double)np_[3] * np_[3];
  const double ninv = (nin2 > 0.0) ? 1.0 / sqrt(nin2) : 1.0;
  const double proj = (nin2 > 0.0) ? (n[0] * g_nh[0] + n[1] * g_nh[1]) + n[2] * g_nh[2] : 0.0;
  double g_nin[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) g_nin[k] = (g_nh[k] - n[k] * proj) * ninv;
  if (type == SRH_PRIM_TRIANGLE) {
    if (G.face[s]) add3(G.face[s] + 12 * (size_t)li, g_q);
  } else if (G.pos[s]) {
    add3(G.pos[s] + 4 * (size_t)li, g_q);
  }
  if (G.normal[s]) add3(G.normal[s] + 4 * (size_t)li, g_nin);
}

}  // namespace srh

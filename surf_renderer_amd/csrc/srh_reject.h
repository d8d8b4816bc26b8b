// Screen-space reject records of the FAST / BINNED render modes (gfx950).
//
// For a pinhole camera the set of pixels whose ray *line* meets a disc (or a sphere) is the interior of
// a conic in pixel coordinates, the inside of a triangle is the intersection of three half planes, and
// n^.D is affine in the pixel coordinates.  k_prep derives those screen-space shapes per primitive and per
// frame in fp64, inflates them by a margin that covers every fp32 rounding on the evaluation side, and
// stores them as fp32 "reject records".  The render kernels evaluate the cheap fp32 tests per
// (pixel, primitive) pair and send only the survivors to the fp64 intersection of srh_device.h.  A pair
// that fails a reject test is provably not the winner of the fp64 path, so every mode's output is
// bit-identical to the all-pairs fp64 mode.
//
// Pixel coordinates are (c, r) = (column, row) indices of the full image; the un-normalised ray
// direction is affine in them:  D(c, r) = D0 + c*Dc + r*Dr   (numpy/renderer.py:152-164).
//
// rec32 layouts (floats)
//   disc (12)      [0] c0 [1] r0 [2] A11 [3] 2*A12 [4] A22 [8], [11] 0 : candidate iff
//                  dc*(A11*dc + 2A12*dr) + A22*dr*dr - 1 <= 0,  dc = c - c0, dr = r - r0
//                  (A = 0: "always a candidate" -- the disc's image is not an ellipse, etc.)
//                  [5..7] u0 u1 u2 [9] lo_u [10] hi_u : depth estimate, see plane_estimate_record
//   sphere (12)    [0..4] as the disc; [5] upper bound of 1 / t for the whole sphere (1e30: none)
//   triangle (16)  {a_i, b_i, g_i} at [4i..4i+2], i < 3: candidate iff min_i(a_i*c + b_i*r + g_i) >= 0
//                  [3] u0 [7] u1 [11] u2 [13] lo_u [14] hi_u
//   plane (8)      [0..2] u0 u1 u2 [4] lo_u [5] hi_u; candidate iff den > hi_u
//
// Depth estimate of a planar primitive (disc, triangle, plane): the ray distance is
//   t = k |D| / (n^.D),   n^.D = v0 + c*v1 + r*v2   with v_j = n^.{D0, Dc, Dr}   (affine in the pixel coordinates)
#pragma once
#include "srh_device.h"

namespace srh {

constexpr int kRec32Stride[4] = {12, 8, 12, 16};

struct PixelBasis {   // D(c,r) = D0 + c*Dc + r*Dr
  double D0[3], Dc[3], Dr[3];
};

__device__ __host__ inline PixelBasis pixel_basis(const FrameDev& F) {
  PixelBasis B;
  for (int i = 0; i < 3; ++i) {
    B.D0[i] = (-F.half_w * F.bx[i] + F.half_h * F.by[i]) - F.focal * F.bz[i];
    B.Dc[i] = (F.step_x * F.half_w) * F.bx[i];
    B.Dr[i] = (F.step_y * F.half_h) * F.by[i];
  }
  return B;
}

__device__ inline double abs_dot3(const double* a, const double* b) {
  return fabs(a[0] * b[0]) + fabs(a[1] * b[1]) + fabs(a[2] * b[2]);
}

__device__ inline void rec_zero(float* out, int n) {
  for (int i = 0; i < n; ++i) out[i] = 0.0f;
}

// Depth-estimate fields of a planar primitive with unit normal n and plane offset k (see the file header):
//   u_j = w_j / K  with  w0 = s v0 + E, w1 = s v1, w2 = s v2,  s = sign(k),  K = |k| (1 - 2^-20),  and lo_u, hi_u.
// The kernel evaluates den = u0 + c u1 + r u2 = (s (n^.D) + E) / K up to an fp32 error below E / K, where
//   E = 2^-19 (|v0| + W |v1| + H |v2|)  covers the three rounded coefficients and the two fmas of the vector path
// (5 x 2^-24) and, with 2x slack, the matrix path (sweep_bin_mfma): the constant moved to the tile origin by two more
// fmas, every coefficient split into three bfloat16 parts (2^-24 of itself left over) and nine exact products summed
// in fp32 (at most 16 x 2^-24 together) -- so s (n^.D) / K <= den always.  A valid hit with near > 0 has s (n^.D) > 0 and ray distance
//   t = |k| |D| / (s n^.D) >= |D| / ((1 - 2^-20) max(den, lo_u))      for ANY lo_u > 0
// (the 2^-20 absorbs the roundings of |D|, the reciprocal and the product).  The kernel uses exactly that,
//   lower bound = |D| rcp(max(den, lo_u)),   lo_u = 1025 E / K:
// clamping at lo_u keeps the bound finite and positive where the plane is seen edge-on (it is then simply weak).
//   den <= hi_u = -1023 E / K   provably t < 0: never a valid hit when near > 0 (used to cull tiles, and per pixel
//                               for planes, whose every pixel is otherwise a candidate)
// An estimate that cannot be represented (plane through the eye, overflow) is WITHDRAWN: u0 = 1e30, u1 = u2 = 0
// make den huge everywhere, i.e. "as near as can be", which ranks the candidate first -- it is then confirmed,
// not trusted.
__device__ inline void plane_estimate_record(const double n[3], double k, const PixelBasis& B, int W, int H,
                                             float* u0, float* u1, float* u2, float* unused, float* lo_u,
                                             float* hi_u) {
  const double a = dot3(n, B.D0), b = dot3(n, B.Dc), c = dot3(n, B.Dr);
  const double sg = (k > 0.0) ? 1.0 : ((k < 0.0) ? -1.0 : 0.0);
  // fp64 side: a, b, c carry 2^-52 of the ABSOLUTE terms they were summed from, and so does the n^.d the exact path
  // divides k by (hit_*64) -- where the plane is seen edge-on over the whole image these exceed 2^-21 of the values
  // themselves, so the slack also holds 2^-48 of the absolute terms (32x the rounding of either side)
  const double e = 1.9073486328125e-6 * (fabs(a) + fabs(b) * W + fabs(c) * H) +
                   3.552713678800501e-15 * (abs_dot3(n, B.D0) + abs_dot3(n, B.Dc) * W + abs_dot3(n, B.Dr) * H);
  const double K = fabs(k) * (1.0 - 9.5367431640625e-7);
  const double big = (fabs(a) + e + fabs(b) * W + fabs(c) * H) / K, lo = 1025.0 * e / K;
  *unused = 0.0f;
  if (!(K > 0.0) || !isfinite(big) || !(big < 1.0e30) || !(lo > 1.0e-30)) {
    *u0 = 1.0e30f;
    *u1 = *u2 = 0.0f;
    *lo_u = 3.0e38f;
    *hi_u = -3.0e38f;
    return;
  }
  *u0 = (float)((sg * a + e) / K); *u1 = (float)(sg * b / K); *u2 = (float)(sg * c / K);
  *lo_u = (float)lo * 1.0000002f;
  *hi_u = (float)(-1023.0 * e / K) * 1.0000002f;
}

// Conic x^T T x <= 0, x = (1, c, r)  ->  inflated, normalised ellipse record out[0..4] (out[11] = 0: the quadratic form
//   dc (A11 dc + 2A12 dr) + A22 dr^2 - 1 <= 0,  dc = c - c0, dr = r - r0).
// Returns the major semi-axis in pixels, or -1 (record = "always a candidate") when the conic is not a
// well-conditioned ellipse: not positive definite, centre or size beyond 2^20 pixels, or axis ratio > 512
// (a disc seen edge-on; its parameters are then numerical noise).
//
// ONE form for every ellipse (until round 3 elongated ones were stored in a principal-axes form).  The record is
// evaluated in two ways, and the inflation `thr` covers both:
//   vector path   centre-relative in fp32 (pair_bounds, ellipse_reject): three rounded coefficients and five rounded
//                 operations on terms up to Tm near the boundary;
//   matrix path   (sweep_bin_mfma) as the polynomial a0 + a1 x + a2 y + a3 x^2 + a4 x y + a5 y^2 in TILE-LOCAL pixel
//                 coordinates x, y in [0, 15]: coefficients moved to the tile origin in fp32 (six rounded operations),
//                 each split into two float16 parts (2^-22 of itself left over), twelve exact products summed in fp32.
//                 Its terms reach the form's value, with absolute values, 15 pixels beyond the tile origin; a binned
//                 tile starts at most 17 pixels outside the ellipse's box.
// Tm = A11 X^2 + 2|A12| X Y + A22 Y^2 + 1 with X = hc + 34, Y = hr + 34 (hc, hr = half extents of the box) bounds the
// sum of the absolute terms of either evaluation; each commits at most 2.6 * 2^-20 Tm, `rel` is 2^-18 Tm.  A point
// inside the true ellipse has A d.d <= 1, hence value <= 1 / thr - 1 <= -rel / (1 + rel) after the inflation.
__device__ inline double conic_record(double T00, double T01, double T02, double T11, double T12, double T22,
                                      int W, int H, float* out) {
  rec_zero(out, 5);
  out[11] = 0.0f;
  const double det = T11 * T22 - T12 * T12;
  if (!(T11 > 0.0) || !(det > 0.0) || !isfinite(T00 + T01 + T02 + T11 + T12 + T22)) return -1.0;
  const double c0 = -(T22 * T01 - T12 * T02) / det;
  const double r0 = -(T11 * T02 - T12 * T01) / det;
  const double F0 = T00 + T01 * c0 + T02 * r0;                 // value at the centre, < 0 inside
  if (!(F0 < 0.0)) return -1.0;
  const double iF = -1.0 / F0;
  const double A11 = T11 * iF, A12 = T12 * iF, A22 = T22 * iF;
  const double mean = 0.5 * (A11 + A22), dev = sqrt(0.25 * (A11 - A22) * (A11 - A22) + A12 * A12);
  const double lmax = mean + dev, lmin = mean - dev;
  if (!(lmin > 0.0) || !isfinite(lmax)) return -1.0;
  const double smin = 1.0 / sqrt(lmax);
  const double smax = 1.0 / sqrt(lmin);
  const double far_lim = 1048576.0;
  if (!(fabs(c0) < far_lim) || !(fabs(r0) < far_lim) || !(smax < far_lim)) return -1.0;
  if (!(smax <= 512.0 * smin)) return -1.0;
  const double dA = A11 * A22 - A12 * A12;
  if (!(dA > 0.0)) return -1.0;
  const double idA = 1.0 / dA;
  const double X = sqrt(A22 * idA) + 34.0, Y = sqrt(A11 * idA) + 34.0;
  const double Tm = A11 * X * X + 2.0 * fabs(A12) * X * Y + A22 * Y * Y + 1.0;
  // Position: c0, r0 are rounded to fp32 (2^-24 |c0|) and so is c - c0 (2^-24 (W + |c0|)), together
  // < 2^-23 (|c0| + |r0| + W + H) pixels; delta doubles that and adds 2^-12 px.
  const double pos_err = 2.384185791015625e-7 * (fabs(c0) + fabs(r0) + W + H);
  const double delta = 0.000244140625 + pos_err;
  const double rel = 3.814697265625e-6 * Tm;
  const double grow = 1.0 + delta / smin;
  const double thr = grow * grow * (1.0 + rel);
  if (!(thr < 1.0e6)) return -1.0;
  const double it = 1.0 / thr;
  out[0] = (float)c0;
  out[1] = (float)r0;
  out[2] = (float)(A11 * it);
  out[3] = (float)(2.0 * A12 * it);
  out[4] = (float)(A22 * it);
  return smax * sqrt(thr);
}

// Upper bound of 1 / t for every hit inside the ball of radius r around the point at eye-relative position -oc:
// t >= |oc| - r (the ball's nearest point).  1e30 ("no estimate") when the eye is inside or on the ball.
// fp64 side: |oc| - r cancels when the eye is close to the ball's surface relative to its size, and the exact path's
// own distance (q = oc + t d against r^2, or the sphere's root -b - sqrt(b^2 - 4 a c)) is then off by up to
// 2^-24 (|oc| + r) (square root of 2^-51 of the squared terms): the bound gives up 2^-22 (|oc| + r), four times that.
__device__ inline float ball_inverse_depth_bound(const double oc[3], double r) {
  const double l = sqrt(dot3(oc, oc));
  const double tmin = (l - r) - 2.384185791015625e-7 * (l + fabs(r));
  if (!(tmin > 0.0) || !isfinite(tmin)) return 1.0e30f;
  const double inv = (1.0 + 9.5367431640625e-7) / tmin;
  return (inv < 1.0e30) ? (float)inv * 1.0000002f : 1.0e30f;
}

// Can the six fp64 coefficients of a conic be trusted?  `tmax` = largest |T_ij|, `emax` = largest sum of the absolute
// values of the terms T_ij was formed from (before they cancelled): the evaluation error is a few 2^-53 emax, and the
// margins of conic_record cover coefficient errors of 2^-26 of the conic's scale -- so trust needs tmax > 2^-24 emax.
// The entries live on different scales (x = (1, c, r): T_ij carries |Pi| |Pj|, and |Dc| is a pixel pitch), so both are
// taken in units of |Pi| |Pj|.
// (A disc with centre y = -1e20 and radius 1e20 passes through the scene, |oc|^2 and r^2 agree to the last bit, and
// the coefficients are rounding noise that happened to look like a small ellipse; found by the non-finite fuzz with
// seed 2002.)  Not trusted -> the record stays "always a candidate".
__device__ inline bool conic_trusted(double tmax, double emax) {
  return isfinite(emax) && tmax * 16777216.0 > emax;
}

// sphere silhouette  cq (Pi.Pj) - (oc.Pi)(oc.Pj)  with the trust test; `sq` = magnitude of the terms cq was formed from
__device__ inline void sphere_conic_record(const double oc[3], double cq, double sq, const double* const P[3], int W, int H,
                                           float* out) {
  double w[3], mw[3], is[3];
  for (int j = 0; j < 3; ++j) {
    w[j] = dot3(oc, P[j]); mw[j] = abs_dot3(oc, P[j]);
    const double l = sqrt(dot3(P[j], P[j]));
    is[j] = (l > 0.0) ? 1.0 / l : 0.0;                    // entries are compared in units of |Pi| |Pj| (see below)
  }
  double t[6], tmax = 0.0, emax = 0.0;
  int q = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 3; ++j, ++q) {
      t[q] = cq * dot3(P[i], P[j]) - w[i] * w[j];
      tmax = fmax(tmax, fabs(t[q]) * (is[i] * is[j]));
      emax = fmax(emax, (sq * abs_dot3(P[i], P[j]) + mw[i] * fabs(w[j]) + mw[j] * fabs(w[i]) + fabs(w[i] * w[j])) *
                            (is[i] * is[j]));
    }
  if (!conic_trusted(tmax, emax)) { rec_zero(out, 5); out[11] = 0.0f; return; }
  (void)conic_record(t[0], t[1], t[2], t[3], t[4], t[5], W, H, out);
}

// disc: | oc (n.D) + k D |^2 <= r^2 (n.D)^2   (numpy/renderer.py:69,85-88 with t = k / (n.D))
__device__ inline void disk_reject_record(const double* R, const double o[3], const PixelBasis& B, int W, int H,
                                          float* out) {
  const double* n = R;
  const double k = R[3];
  const double oc[3] = {o[0] - R[4], o[1] - R[5], o[2] - R[6]};
  const double r2 = R[7];
  const double* P[3] = {B.D0, B.Dc, B.Dr};
  // mu: |terms of u| summed over the components, au: |u| (L1).  The exact path forms q = (eye + t d) - c
  // (hit_disk64, the reference's order): its eye + t d carries 2^-53 (|eye| + |t d|), which in the units of u = q (n.D)
  // is |eye| |n.P| + |k| |P| -- so |eye| joins the terms, and a camera 1e10 away from the origin makes the record
  // untrusted instead of wrong
  double nu[3], u[3][3], mu[3], au[3], is[3];
  const double ol1 = fabs(o[0]) + fabs(o[1]) + fabs(o[2]);
  for (int j = 0; j < 3; ++j) {
    nu[j] = dot3(n, P[j]);
    mu[j] = au[j] = 0.0;
    const double l = sqrt(dot3(P[j], P[j]));
    is[j] = (l > 0.0) ? 1.0 / l : 0.0;
    for (int a = 0; a < 3; ++a) {
      u[j][a] = oc[a] * nu[j] + k * P[j][a];
      mu[j] += fabs(oc[a] * nu[j]) + fabs(k * P[j][a]);
      au[j] += fabs(u[j][a]);
    }
    mu[j] += ol1 * fabs(nu[j]);
  }
  double t[6], tmax = 0.0, emax = 0.0;
  {
    int q = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = i; j < 3; ++j, ++q) {
        t[q] = dot3(u[i], u[j]) - r2 * nu[i] * nu[j];
        tmax = fmax(tmax, fabs(t[q]) * (is[i] * is[j]));
        emax = fmax(emax, (mu[i] * au[j] + mu[j] * au[i] + abs_dot3(u[i], u[j]) + fabs(r2 * nu[i] * nu[j])) *
                              (is[i] * is[j]));
      }
  }
  bool degenerate = true;
  if (conic_trusted(tmax, emax)) degenerate = conic_record(t[0], t[1], t[2], t[3], t[4], t[5], W, H, out) < 0.0;
  if (degenerate) {
    // The disc's own image is not a usable ellipse (seen edge-on: axis ratio beyond 512, parameters are noise).
    // Every hit lies on the disc, hence inside the sphere around its centre with its radius, and that sphere's
    // silhouette is a robust, well-conditioned ellipse.  Rare (|cos| < 2e-3), so the extra work is off the common path.
    const double oo = dot3(oc, oc);
    sphere_conic_record(oc, oo - r2, oo + fabs(r2) + 2.0 * sqrt(dot3(o, o) * oo), P, W, H, out);
  }
  plane_estimate_record(n, k, B, W, H, out + 5, out + 6, out + 7, out + 8, out + 9, out + 10);
  // A stand-in shape passes pixels the disc does not cover, and the plane-distance estimate means nothing there:
  // replace it by the bounding ball's (every hit lies in the ball, so t >= |oc| - r).  The kernel multiplies den by
  // 1 / |D|, so the constant is scaled by the largest |D| of the image (|D| is convex: a corner) -- a little weak, valid.
  if (degenerate) {
    double dmax = 0.0;
    for (int i = 0; i < 4; ++i) {
      const double c = (i & 1) ? (double)(W - 1) : 0.0, rr = (i & 2) ? (double)(H - 1) : 0.0;
      const double D[3] = {B.D0[0] + c * B.Dc[0] + rr * B.Dr[0], B.D0[1] + c * B.Dc[1] + rr * B.Dr[1],
                           B.D0[2] + c * B.Dc[2] + rr * B.Dr[2]};
      dmax = fmax(dmax, sqrt(dot3(D, D)));
    }
    const double u = (double)ball_inverse_depth_bound(oc, sqrt(r2)) * dmax * 1.000001;
    out[5] = (u < 1.0e30 && isfinite(u)) ? (float)u * 1.0000002f : 1.0e30f;
    out[6] = out[7] = 0.0f; out[9] = 3.0e38f; out[10] = -3.0e38f;
  }
}

// sphere: the ray's line meets it iff (oc.D)^2 - |D|^2 (|oc|^2 - r^2) >= 0   (numpy/renderer.py:20-25).
// With near <= 0 a missed line yields the valid t = 0 (Q2), so nothing may be rejected.
__device__ inline void sphere_reject_record(const double* R, const PixelBasis& B, int W, int H, bool near_positive,
                                            bool tch, float* out) {
  rec_zero(out, 12);
  out[5] = 1.0e30f;
  if (!near_positive) return;
  const double* oc = R;
  const double cq = R[3];
  const double* P[3] = {B.D0, B.Dc, B.Dr};
  double w[3];
  for (int j = 0; j < 3; ++j) w[j] = dot3(oc, P[j]);
  // cq = |oc|^2 - r^2 comes from the exact record: the terms it was formed from are at most 2 |oc|^2 + |cq|
  sphere_conic_record(oc, cq, 2.0 * dot3(oc, oc) + fabs(cq), P, W, H, out);
  // Depth bound.  Eye outside the sphere (cq > 0): both roots have the sign of (pos - eye).D = -oc.D.  Positive roots
  // give t = t1 >= |oc| - r.  Negative roots are a miss under the torch semantics, but the numpy backend's sentinel
  // turns them into t = 1.0 (Q2) -- so there the bound holds only if -oc.D > 0 on the whole image (affine: its
  // minimum is at a corner); otherwise it is capped at t >= min(|oc| - r, 1).
  if (cq > 0.0) {
    const double r = sqrt(fmax(dot3(oc, oc) - cq, 0.0));
    float inv = ball_inverse_depth_bound(oc, r);
    if (!tch) {
      double front = 1e300;
      for (int i = 0; i < 4; ++i) {
        const double c = (i & 1) ? (double)(W - 1) : 0.0, rr = (i & 2) ? (double)(H - 1) : 0.0;
        front = fmin(front, -(w[0] + c * w[1] + rr * w[2]));
      }
      if (!(front > 0.0)) inv = fmaxf(inv, 1.000001f);
    }
    out[5] = inv;
  }
}

// triangle: for a valid hit (t >= near > 0) sign(n.D) = sign(k), so edge i is satisfied iff
// sign(k) * (s_i n + k m_i) . D >= 0 with m_i = n x e_i, s_i = (o - v_i) . m_i   (numpy/renderer.py:114-126)
//
// fp64 side.  E_i(D) = sg (s_i n + k m_i) . D is a scalar triple product expanded, and it cancels when vertices lie far
// away on two axes (terms |o - v| |e| ~ 1e40 for a result ~ 1e20): the coefficients computed here, and the value
// dot(cross(e_i, p - v_i), n^) the exact path computes per pixel, then both differ from E_i by rounding noise that has
// nothing to do with the fp32 margin.  Both are bounded through the ABSOLUTE terms: with am_j = |n_a e_b| + |n_b e_a|
// (terms of m_j), as = sum_j (|o_j| + |v_j|) am_j (terms of s) and Dsum >= every |D_j| of the image,
//   |computed E_i - E_i|,  |exact path's value * (n^.D) - E_i|   <=   2^-48 (2 as + |k| sum_j am_j) Dsum  =: err
// (32x the 2^-53 of each rounding; the exact path's own terms are the same ones: its p - v_i carries
// 2^-53 (|o| + |v| + |t d|), its n^.d another 2^-51, which multiplies s_i).  `err`, in pixels after the
// normalisation, is added to the edge's margin; an edge whose noise reaches a quarter pixel stays "always a candidate".
__device__ inline void triangle_reject_record(const double* R, const double o[3], const PixelBasis& B, int W, int H,
                                              bool near_positive, float* out) {
  rec_zero(out, 16);
  for (int i = 0; i < 3; ++i) out[4 * i + 2] = 1.0f;          // a = b = 0, g = 1: always a candidate
  const double* n = R;
  const double k = R[3];
  plane_estimate_record(n, k, B, W, H, out + 3, out + 7, out + 11, out + 12, out + 13, out + 14);
  if (!near_positive) return;
  const double sg = (k > 0.0) ? 1.0 : ((k < 0.0) ? -1.0 : 0.0);
  double dsum = 0.0;
  for (int j = 0; j < 3; ++j) dsum += fabs(B.D0[j]) + fabs(B.Dc[j]) * W + fabs(B.Dr[j]) * H;
  for (int i = 0; i < 3; ++i) {
    const double* v = R + 4 + 3 * i;
    const double* e = R + 13 + 3 * i;
    const double m[3] = {n[1] * e[2] - n[2] * e[1], n[2] * e[0] - n[0] * e[2], n[0] * e[1] - n[1] * e[0]};
    const double am[3] = {fabs(n[1] * e[2]) + fabs(n[2] * e[1]), fabs(n[2] * e[0]) + fabs(n[0] * e[2]),
                          fabs(n[0] * e[1]) + fabs(n[1] * e[0])};
    const double ov[3] = {o[0] - v[0], o[1] - v[1], o[2] - v[2]};
    const double s = dot3(ov, m);
    const double as = (fabs(o[0]) + fabs(v[0])) * am[0] + (fabs(o[1]) + fabs(v[1])) * am[1] + (fabs(o[2]) + fabs(v[2])) * am[2];
    const double w[3] = {sg * (s * n[0] + k * m[0]), sg * (s * n[1] + k * m[1]), sg * (s * n[2] + k * m[2])};
    double a = dot3(w, B.Dc), b = dot3(w, B.Dr), g = dot3(w, B.D0);
    const double len = sqrt(a * a + b * b);
    if (!(len > 0.0) || !isfinite(len) || !isfinite(g)) continue;
    const double err = 3.552713678800501e-15 * (2.0 * as + fabs(k) * (am[0] + am[1] + am[2])) * dsum / len;   // pixels
    if (!(err < 0.25)) continue;
    a /= len; b /= len; g /= len;                             // signed distance to the edge line, pixels
    if (!(fabs(g) < 1.0e9)) continue;
    // fp32 evaluation of a*c + b*r + g with rounded a, b, g: error < 3 * 2^-24 (|g| + W + H); margin 2.7x that
    g += 0.000244140625 + 4.76837158203125e-7 * (fabs(g) + W + H) + err;
    out[4 * i + 0] = (float)a;
    out[4 * i + 1] = (float)b;
    out[4 * i + 2] = (float)g;
  }
}

__device__ inline void plane_reject_record(const double* R, const PixelBasis& B, int W, int H, float* out) {
  rec_zero(out, 8);
  plane_estimate_record(R, R[3], B, W, H, out + 0, out + 1, out + 2, out + 3, out + 4, out + 5);
}

// fp32 evaluation of the ellipse record for N pixels of one row: value <= 0 means "candidate".
template <int N>
__device__ __forceinline__ void ellipse_reject(const float* __restrict__ R, const float (&cf)[N], float rf,
                                               float (&q)[N]) {
  const float dr = rf - R[1];
  const float ee = R[3] * dr;
  const float gg = __builtin_fmaf(R[4] * dr, dr, -1.0f);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float dc = cf[j] - R[0];
    q[j] = __builtin_fmaf(dc, __builtin_fmaf(R[2], dc, ee), gg);
  }
}

// ---- conservative pixel bounding boxes of the reject shapes (for tile binning) ------------------------
struct BBox {
  double c0, c1, r0, r1;   // inclusive pixel range that contains every true hit of the primitive
  bool full;               // no useful bound: treat as covering the whole image
};

__device__ inline BBox bbox_full() { return BBox{0, 0, 0, 0, true}; }

// Box of the stored ellipse record (either form), one pixel of slack.
__device__ inline BBox conic_bbox(const float* rec) {
  const double A11 = rec[2], A12 = 0.5 * (double)rec[3], A22 = rec[4];
  const double det = A11 * A22 - A12 * A12;
  if (!(A11 > 0.0) || !(det > 0.0)) return bbox_full();
  const double hc = sqrt(A22 / det), hr = sqrt(A11 / det);
  if (!isfinite(hc) || !isfinite(hr)) return bbox_full();
  return BBox{(double)rec[0] - hc - 1.0, (double)rec[0] + hc + 1.0, (double)rec[1] - hr - 1.0,
              (double)rec[1] + hr + 1.0, false};
}

// Box of the triangle cut out by the three stored (inflated) edge half-planes, if it is bounded.
// Three half-planes a_i c + b_i r + g_i >= 0 with unit normals bound a region iff the normals turn the same way round
// from each to the next (all three cross products of one sign): otherwise some direction leaves none of them, and the
// region is a wedge, a strip or a half-plane.  The latter is what a huge triangle seen from near its plane looks like:
// its far edges all project onto the plane's HORIZON, three nearly identical lines -- whose pairwise "intersections"
// are rounding noise that used to pass for a tiny triangle somewhere off the image, and the primitive was then not
// binned at all (found by the adversarial campaign, seed 3505 scene 2238; tests/test_hip_adversarial.py).  Cross
// products below 1e-5 are refused too: a corner is (g_i n_j - g_j n_i) / D with |g| up to 1e9 pixels.
__device__ inline BBox triangle_bbox(const float* rec) {
  double a[3], b[3], g[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = rec[4 * i]; b[i] = rec[4 * i + 1]; g[i] = rec[4 * i + 2];
    if (a[i] == 0.0 && b[i] == 0.0) return bbox_full();
  }
  const double D01 = a[0] * b[1] - a[1] * b[0], D12 = a[1] * b[2] - a[2] * b[1], D20 = a[2] * b[0] - a[0] * b[2];
  const bool ccw = D01 > 1.0e-5 && D12 > 1.0e-5 && D20 > 1.0e-5, cw = D01 < -1.0e-5 && D12 < -1.0e-5 && D20 < -1.0e-5;
  if (!ccw && !cw) return bbox_full();
  double cmin = 1e300, cmax = -1e300, rmin = 1e300, rmax = -1e300;
  for (int i = 0; i < 3; ++i) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const double D = a[i] * b[j] - a[j] * b[i];
    const double c = (-g[i] * b[j] + g[j] * b[i]) / D;
    const double r = (-a[i] * g[j] + a[j] * g[i]) / D;
    if (!isfinite(c) || !isfinite(r)) return bbox_full();
    if (!(a[k] * c + b[k] * r + g[k] >= -1e-6 * (fabs(c) + fabs(r) + fabs(g[k]) + 1.0))) return bbox_full();
    cmin = fmin(cmin, c); cmax = fmax(cmax, c); rmin = fmin(rmin, r); rmax = fmax(rmax, r);
  }
  // the corners carry the rounding of the division: 2^-52 (|g_i| + |g_j|) / |D| pixels, far below the pixel of slack
  // for |g| < 1e9 and |D| > 1e-5 only if |g| stays below ~1e6 -- beyond that the slack grows with it
  const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  const double slack = 1.0 + 1.0e-10 * gmax;
  return BBox{cmin - slack, cmax + slack, rmin - slack, rmax + slack, false};
}

// Can a pixel of the rectangle [c0,c1] x [r0,r1] (inclusive pixel coordinates) be a valid hit of the primitive whose
// reject record this is?  Used to drop tiles of a primitive's box that it cannot reach (box corners of round or
// slanted shapes, tiles where its plane is behind the eye).  Two proofs of "no":
//   shape   true hits lie inside the stored (inflated) reject shape, so the exact fp64 minimum of the stored
//           ellipse form over the rectangle being > 1, or one stored triangle edge function being negative at all
//           four corners, rules the rectangle out;
//   plane   den is affine, so its maximum sits at a corner; evaluated in fp64 on the stored coefficients it is
//           >= the true scaled s (n^.D), and a value <= hi_u means t < 0 everywhere (only used when near > 0).
struct RectTest {
  int type;
  bool ellipse, behind;
  double x0, y0, A11, A12, A22, i11, i22;       // ellipse centre, quadratic form, 1 / A11, 1 / A22
  double ea[3], eb[3], eg[3];                   // triangle edge functions
  double u0, u1, u2, hi;                        // depth-estimate denominator

  __device__ inline RectTest(int type_, const float* rec, bool near_positive) : type(type_), ellipse(false), behind(false) {
    if (type == SRH_PRIM_TRIANGLE) {
      for (int i = 0; i < 3; ++i) { ea[i] = rec[4 * i]; eb[i] = rec[4 * i + 1]; eg[i] = rec[4 * i + 2]; }
      u0 = rec[3]; u1 = rec[7]; u2 = rec[11]; hi = rec[14];
      behind = near_positive;
    } else if (type == SRH_PRIM_DISK || type == SRH_PRIM_SPHERE) {
      A11 = rec[2]; A12 = 0.5 * (double)rec[3]; A22 = rec[4];
      ellipse = (A11 > 0.0) && (A22 > 0.0);                    // else "always a candidate"
      i11 = ellipse ? 1.0 / A11 : 0.0;
      i22 = ellipse ? 1.0 / A22 : 0.0;
      x0 = rec[0]; y0 = rec[1];
      if (type == SRH_PRIM_DISK) { u0 = rec[5]; u1 = rec[6]; u2 = rec[7]; hi = rec[10]; behind = near_positive; }
    }
  }

  __device__ inline bool reaches(double c0, double c1, double r0, double r1) const {
    if (behind && u0 + fmax(u1 * c0, u1 * c1) + fmax(u2 * r0, u2 * r1) <= hi) return false;
    if (type == SRH_PRIM_TRIANGLE) {
      for (int i = 0; i < 3; ++i)
        if (ea[i] * (ea[i] > 0.0 ? c1 : c0) + eb[i] * (eb[i] > 0.0 ? r1 : r0) + eg[i] < 0.0) return false;
      return true;
    }
    if (!ellipse) return true;
    if (x0 >= c0 && x0 <= c1 && y0 >= r0 && y0 <= r1) return true;
    // centre outside: the minimum of the form over the rectangle is on its boundary; per edge the form is a
    // parabola in the free coordinate, minimised at its clamped vertex
    double qmin = 1e300;
    for (int e = 0; e < 2; ++e) {
      const double dc = (e ? c1 : c0) - x0;
      const double dr = fmin(fmax(-A12 * dc * i22, r0 - y0), r1 - y0);
      qmin = fmin(qmin, A11 * dc * dc + 2.0 * A12 * dc * dr + A22 * dr * dr);
      const double dr2 = (e ? r1 : r0) - y0;
      const double dc2 = fmin(fmax(-A12 * dr2 * i11, c0 - x0), c1 - x0);
      qmin = fmin(qmin, A11 * dc2 * dc2 + 2.0 * A12 * dc2 * dr2 + A22 * dr2 * dr2);
    }
    return !(qmin > 1.000001);
  }
};

}  // namespace srh

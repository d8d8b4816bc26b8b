// Conservative per-pair reject tests of the FAST / binned render modes (gfx950).
//
// For a pinhole camera the set of pixels whose ray *line* meets a disc (or a sphere) is the interior of
// a conic in pixel coordinates, and the inside of a triangle is the intersection of three half planes.
// k_prep derives those screen-space shapes per primitive and per frame in fp64, inflates them by a
// margin that covers every fp32 rounding on the evaluation side, and stores them as fp32 "reject
// records".  The render kernels evaluate the cheap fp32 test per (pixel, primitive) pair and send only
// the survivors to the fp64 intersection of srh_device.h.  A pair that fails the reject test is provably
// a miss of the fp64 test, so the output is bit-identical to the all-pairs fp64 mode.
//
// Pixel coordinates are (c, r) = (column, row) indices of the full image; the un-normalised ray
// direction is affine in them:  D(c, r) = D0 + c*Dc + r*Dr   (numpy/renderer.py:152-164).
//
// rec32 layouts (floats)
//   disc, sphere [0] c0 [1] r0 [2] A11 [3] 2*A12 [4] A22  -> candidate iff
//                dc*(A11*dc + 2A12*dr) + A22*dr*dr - 1 <= 0,  dc = c - c0, dr = r - r0
//                (all zero = "always a candidate": discs whose image is not an ellipse, etc.)
//   triangle     3 x {a, b, g, 0}: candidate iff min_i(a_i*c + b_i*r + g_i) >= 0
//   plane        none: every pixel is a candidate
#pragma once
#include "srh_device.h"

namespace srh {

constexpr int kRec32Stride[4] = {8, 4, 8, 12};

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

__device__ inline void rec_always(float* out, int n) {
  for (int i = 0; i < n; ++i) out[i] = 0.0f;
}

// Conic x^T T x <= 0, x = (1, c, r)  ->  inflated, normalised ellipse record.
__device__ inline void conic_record(double T00, double T01, double T02, double T11, double T12, double T22,
                                    int W, int H, float* out) {
  rec_always(out, 8);
  const double det = T11 * T22 - T12 * T12;
  if (!(T11 > 0.0) || !(det > 0.0) || !isfinite(T00 + T01 + T02 + T11 + T12 + T22)) return;
  const double c0 = -(T22 * T01 - T12 * T02) / det;
  const double r0 = -(T11 * T02 - T12 * T01) / det;
  const double F0 = T00 + T01 * c0 + T02 * r0;                 // value at the centre, < 0 inside
  if (!(F0 < 0.0)) return;
  double A11 = T11 / -F0, A12 = T12 / -F0, A22 = T22 / -F0;
  const double mean = 0.5 * (A11 + A22), dev = sqrt(0.25 * (A11 - A22) * (A11 - A22) + A12 * A12);
  const double lmax = mean + dev, lmin = mean - dev;
  if (!(lmin > 0.0) || !isfinite(lmax)) return;
  double smin = 1.0 / sqrt(lmax);
  const double smax = 1.0 / sqrt(lmin);
  const double far_lim = 1048576.0;
  if (!(fabs(c0) < far_lim) || !(fabs(r0) < far_lim) || !(smax < far_lim)) return;
  if (smax > 32.0 * smin) {            // thin sliver: the quadratic form would cancel badly in fp32
    A11 = A22 = 1.0 / (smax * smax);   // -> bounding circle
    A12 = 0.0;
    smin = smax;
  }
  // margin in pixels: covers fp32 rounding of c0, r0, the coefficients and the evaluation
  const double delta = 0.015625 + 9.5367431640625e-7 * (fabs(c0) + fabs(r0) + W + H);
  const double grow = 1.0 + delta / smin;
  const double thr = grow * grow * (1.0 + 0.001953125);
  out[0] = (float)c0;
  out[1] = (float)r0;
  out[2] = (float)(A11 / thr);
  out[3] = (float)(2.0 * A12 / thr);
  out[4] = (float)(A22 / thr);
}

// disc: | oc (n.D) + k D |^2 <= r^2 (n.D)^2   (numpy/renderer.py:69,85-88 with t = k / (n.D))
__device__ inline void disk_reject_record(const double* R, const PixelBasis& B, int W, int H, float* out) {
  const double* n = R;
  const double k = R[3];
  const double* oc = R + 4;
  const double r2 = R[7];
  const double* P[3] = {B.D0, B.Dc, B.Dr};
  double nu[3], u[3][3];
  for (int j = 0; j < 3; ++j) {
    nu[j] = dot3(n, P[j]);
    for (int a = 0; a < 3; ++a) u[j][a] = oc[a] * nu[j] + k * P[j][a];
  }
  auto T = [&](int i, int j) { return dot3(u[i], u[j]) - r2 * nu[i] * nu[j]; };
  conic_record(T(0, 0), T(0, 1), T(0, 2), T(1, 1), T(1, 2), T(2, 2), W, H, out);
}

// sphere: the ray's line meets it iff (oc.D)^2 - |D|^2 (|oc|^2 - r^2) >= 0   (numpy/renderer.py:20-25).
// With near <= 0 a missed line yields the valid t = 0 (Q2), so nothing may be rejected.
__device__ inline void sphere_reject_record(const double* R, const PixelBasis& B, int W, int H, bool near_positive,
                                            float* out) {
  if (!near_positive) { rec_always(out, 8); return; }
  const double* oc = R;
  const double cq = R[3];
  const double* P[3] = {B.D0, B.Dc, B.Dr};
  double w[3];
  for (int j = 0; j < 3; ++j) w[j] = dot3(oc, P[j]);
  auto T = [&](int i, int j) { return cq * dot3(P[i], P[j]) - w[i] * w[j]; };
  conic_record(T(0, 0), T(0, 1), T(0, 2), T(1, 1), T(1, 2), T(2, 2), W, H, out);
}

// triangle: for a valid hit (t >= near > 0) sign(n.D) = sign(k), so edge i is satisfied iff
// sign(k) * (s_i n + k m_i) . D >= 0 with m_i = n x e_i, s_i = (o - v_i) . m_i   (numpy/renderer.py:114-126)
__device__ inline void triangle_reject_record(const double* R, const double o[3], const PixelBasis& B, int W, int H,
                                              bool near_positive, float* out) {
  rec_always(out, 12);
  for (int i = 0; i < 3; ++i) out[4 * i + 2] = 1.0f;          // a = b = 0, g = 1: always a candidate
  if (!near_positive) return;
  const double* n = R;
  const double k = R[3];
  const double sg = (k > 0.0) ? 1.0 : ((k < 0.0) ? -1.0 : 0.0);
  for (int i = 0; i < 3; ++i) {
    const double* v = R + 4 + 3 * i;
    const double* e = R + 13 + 3 * i;
    const double m[3] = {n[1] * e[2] - n[2] * e[1], n[2] * e[0] - n[0] * e[2], n[0] * e[1] - n[1] * e[0]};
    const double ov[3] = {o[0] - v[0], o[1] - v[1], o[2] - v[2]};
    const double s = dot3(ov, m);
    const double w[3] = {sg * (s * n[0] + k * m[0]), sg * (s * n[1] + k * m[1]), sg * (s * n[2] + k * m[2])};
    double a = dot3(w, B.Dc), b = dot3(w, B.Dr), g = dot3(w, B.D0);
    const double len = sqrt(a * a + b * b);
    if (!(len > 0.0) || !isfinite(len) || !isfinite(g)) continue;
    a /= len; b /= len; g /= len;                             // signed distance to the edge line, pixels
    if (!(fabs(g) < 1.0e9)) continue;
    g += 0.015625 + 9.5367431640625e-7 * (fabs(g) + W + H);
    out[4 * i + 0] = (float)a;
    out[4 * i + 1] = (float)b;
    out[4 * i + 2] = (float)g;
  }
}

}  // namespace srh

// ---- conservative pixel bounding boxes of the reject shapes (for tile binning) ------------------------
namespace srh {

struct BBox {
  double c0, c1, r0, r1;   // inclusive pixel range that contains every true hit of the primitive
  bool full;               // no useful bound: treat as covering the whole image
};

__device__ inline BBox bbox_full() { return BBox{0, 0, 0, 0, true}; }

// Box of { dc*(A11*dc + B*dr) + A22*dr^2 <= 1 } from the *stored* record, one pixel of slack.
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
__device__ inline BBox triangle_bbox(const float* rec) {
  double a[3], b[3], g[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = rec[4 * i]; b[i] = rec[4 * i + 1]; g[i] = rec[4 * i + 2];
    if (a[i] == 0.0 && b[i] == 0.0) return bbox_full();
  }
  double cmin = 1e300, cmax = -1e300, rmin = 1e300, rmax = -1e300;
  for (int i = 0; i < 3; ++i) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const double D = a[i] * b[j] - a[j] * b[i];
    if (!(fabs(D) > 1e-9)) return bbox_full();
    const double c = (-g[i] * b[j] + g[j] * b[i]) / D;
    const double r = (-a[i] * g[j] + a[j] * g[i]) / D;
    if (!isfinite(c) || !isfinite(r)) return bbox_full();
    if (!(a[k] * c + b[k] * r + g[k] >= -1e-6 * (fabs(c) + fabs(r) + fabs(g[k]) + 1.0))) return bbox_full();
    cmin = fmin(cmin, c); cmax = fmax(cmax, c); rmin = fmin(rmin, r); rmax = fmax(rmax, r);
  }
  return BBox{cmin - 1.0, cmax + 1.0, rmin - 1.0, rmax + 1.0, false};
}

}  // namespace srh

// BINNED mode kernels (gfx950): screen-space tile binning + one WAVE per 16x16-pixel tile.
//
//   k_prep (srh.hip)   also writes each primitive's tile range and either counts it into the bins it
//                      overlaps (atomicAdd on per-bin counters) or appends it to its batch's `large` list
//   k_bin_scan         exclusive prefix sum of the per-bin counts (one workgroup)
//   k_bin_fill         second pass over the primitives: claim a slot per overlapped bin, write the index
//   k_render_binned    per tile: sweep the tile's primitives, confirm the front one per pixel, shade, store
//
// A bin is (object batch, tile): every list the render kernel walks holds primitives of ONE type, so its
// loops are specialised per type and the primitive stream is wave-uniform -- list entries and reject records
// arrive through scalar loads into SGPRs, no LDS staging and no barriers.
//
// Bin contents come out of atomics in arbitrary order; the winner is the lexicographic minimum of
// (t, global index), which does not depend on visiting order, so frames are bit-reproducible and
// identical to the ordered all-pairs modes.
#pragma once
#include "srh_device.h"
#include "srh_reject.h"

namespace srh {

__device__ __forceinline__ int segment_of(const FrameDev& F, int gidx) {
  int s = 0;
#pragma unroll
  for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
    if (i < F.nseg && gidx >= F.seg[i].first) s = i;
  return s;
}

// ---- tile range of one primitive (called from k_prep) ------------------------------------------------
__device__ inline void bin_primitive(const FrameDev& F, int seg, int type, const float* rec32, int gidx) {
  BBox b = bbox_full();
  if (type == SRH_PRIM_DISK || type == SRH_PRIM_SPHERE) b = conic_bbox(rec32);
  else if (type == SRH_PRIM_TRIANGLE) b = triangle_bbox(rec32);
  uint16_t* tr = F.tilerange + 4 * (size_t)gidx;
  tr[0] = 1; tr[1] = 0; tr[2] = 0; tr[3] = 0;                    // default: not binned
  bool is_large = b.full;
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  if (!b.full) {
    // inclusive pixel box, clamped to the rendered slab; empty -> the primitive touches no pixel at all
    const double c_lo = fmax(floor(b.c0), 0.0), c_hi = fmin(ceil(b.c1), (double)(F.W - 1));
    const double r_lo = fmax(floor(b.r0), (double)F.row0), r_hi = fmin(ceil(b.r1), (double)(F.row1 - 1));
    if (!(c_lo <= c_hi) || !(r_lo <= r_hi)) return;
    tx0 = (int)c_lo / kTile; tx1 = (int)c_hi / kTile;
    ty0 = ((int)r_lo - F.row0) / kTile; ty1 = ((int)r_hi - F.row0) / kTile;
    is_large = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) > kMaxTilesPerPrim;
  }
  if (is_large) {                                                 // the batch's region of `large` starts at seg.first
    const uint32_t slot = atomicAdd(&F.counters[seg], 1u);
    F.large[F.seg[seg].first + slot] = (uint32_t)gidx;
    return;
  }
  tr[0] = (uint16_t)tx0; tr[1] = (uint16_t)ty0; tr[2] = (uint16_t)tx1; tr[3] = (uint16_t)ty1;
  uint32_t* count = F.counters + kCounterPad + seg * F.ntiles_pad;
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&count[ty * F.tiles_x + tx], 1u);
}

// ---- exclusive scan of the bin counts: one 1024-thread workgroup, 16-byte loads ---------------------------
__global__ __launch_bounds__(1024) void k_bin_scan(FrameDev F) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  const int n4 = F.nbins / 4;                              // nbins is a multiple of 4
  const int per = (n4 + 1023) / 1024;
  const int lo = min(tid * per, n4), hi = min(lo + per, n4);
  const uint4* cnt = reinterpret_cast<const uint4*>(F.counters + kCounterPad);
  uint32_t sum = 0;
  for (int i = lo; i < hi; ++i) {
    const uint4 v = cnt[i];
    sum += (v.x + v.y) + (v.z + v.w);
  }
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {               // Hillis-Steele inclusive scan of the partials
    const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;                          // exclusive prefix of this thread's chunk
  uint4* out = reinterpret_cast<uint4*>(F.tile_off);
  for (int i = lo; i < hi; ++i) {
    const uint4 v = cnt[i];
    uint4 o;
    o.x = run; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
    run = o.w + v.w;
    out[i] = o;
  }
  if (tid == 1023) F.tile_off[F.nbins] = part[1023];
}

// ---- fill: 16 lanes per primitive, one (primitive, tile) pair per lane and step ----------------------------
__global__ __launch_bounds__(256) void k_bin_fill(FrameDev F) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int gidx = t >> 4, sub = t & 15;
  if (gidx >= F.total) return;
  const uint16_t* tr = F.tilerange + 4 * (size_t)gidx;
  const int tx0 = tr[0], ty0 = tr[1], tx1 = tr[2], ty1 = tr[3];
  if (tx0 > tx1) return;
  const int bin0 = segment_of(F, gidx) * F.ntiles_pad;
  uint32_t* cursor = F.counters + kCounterPad + F.nbins;
  const int nx = tx1 - tx0 + 1, n = nx * (ty1 - ty0 + 1);
  for (int k = sub; k < n; k += 16) {
    const int bin = bin0 + (ty0 + k / nx) * F.tiles_x + (tx0 + k % nx);
    const uint32_t slot = atomicAdd(&cursor[bin], 1u);
    F.entries[F.tile_off[bin] + slot] = (uint32_t)gidx;
  }
}

// ---- render: one wave per 16x16-pixel tile, four pixels (one row quad) per lane ---------------------------------
//
// sweep     wave-uniform loop over a list: per (pixel, primitive) pair the fp32 screen-space reject, and for
//           the survivors an fp32 LOWER BOUND of the ray distance, t ~ k |D| / (v0 + c v1 + r v2) minus its error
//           bound.  Per pixel only the smallest bound (with its primitive) and the second smallest are kept.
// confirm   the front candidate of every pixel goes through the fp64 intersection.  If the second smallest bound
//           exceeds that confirmed depth nobody else can win or tie and the pixel is done -- the common case.
// re-sweep  otherwise (a wave-level vote) the lists are walked again and every candidate whose bound still
//           reaches the confirmed depth is confirmed as well.
// A candidate is skipped only when its lower bound exceeds an exactly confirmed depth, so the output equals
// the all-pairs fp64 mode bit for bit.
struct QuadState {
  float cf[4];        // pixel columns as fp32 (exact integers)
  float rf;           // pixel row
  float len[4];       // |D| of the un-normalised ray direction
  float lo1[4];       // smallest depth lower bound seen
  float lo2[4];       // second smallest
  int g1[4];          // global index of the primitive with bound lo1, -1 = none
  float bound[4];     // fp32 value >= the confirmed fp64 depth (re-sweep only)
  bool again[4];      // pixel takes part in the re-sweep
};

// fp32 value that is certainly >= the fp64 depth (the conversion may round down by half an ulp)
__device__ __forceinline__ float float_above(double t) { return (float)t * 1.0000005f; }

__device__ __forceinline__ void confirm_global(const FrameDev& F, int gidx, const double d[3],
                                                         double& best, int& besti) {
  const int s = segment_of(F, gidx);
  int type = F.seg[0].type, first = F.seg[0].first;
  const double* base = F.seg[0].rec64;
#pragma unroll
  for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
    if (s == i) { type = F.seg[i].type; first = F.seg[i].first; base = F.seg[i].rec64; }
  const double* R = base + (size_t)(gidx - first) * kRec64Stride[type];
  resolve_lex(F, hit_any64(type, R, F.o, d), gidx, best, besti);
}

// One reject record in registers: loaded with 16-byte accesses from a wave-uniform address, so the whole
// record arrives through the scalar cache in one clause before any of it is used.
template <int TYPE>
struct RejectRecord {
  static constexpr int kQuads = kRec32Stride[TYPE] / 4;
  float v[kRec32Stride[TYPE]];
  __device__ __forceinline__ void load(const float* __restrict__ src) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int k = 0; k < kQuads; ++k) {
      const float4 t = s4[k];
      v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
    }
  }
  __device__ __forceinline__ float operator[](int i) const { return v[i]; }
};

// Per pixel: is the pair a candidate (passes the screen-space reject), and if so an fp32 lower bound `lo` of
// its ray distance, or `loose` = no usable bound (grazing plane, sphere, near <= 0): must be confirmed.
// Planar depth estimate: t = k |D| / den with den = v0 + c v1 + r v2; E bounds the fp32 error of den, and
// only |den| >= Esolid = 1024 E is trusted, where the relative error of t is below E/|den| + 2^-20.
// t <= 0 with a trusted den is provably not a valid hit (near > 0): the pair is dropped.
template <int TYPE, bool PRETEST>
__device__ __forceinline__ void pair_bounds(const RejectRecord<TYPE>& R, const QuadState& Q, bool cand[4],
                                            bool loose[4], float lo[4]) {
  float den[4], klen[4];
  float E = 0.0f, Esolid = 0.0f;
  if (TYPE == SRH_PRIM_DISK || TYPE == SRH_PRIM_SPHERE) {
    float q[4];
    ellipse_reject4(R.v, Q.cf, Q.rf, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) cand[j] = q[j] <= 0.0f;
    if (TYPE == SRH_PRIM_DISK) {
      const float rowden = __builtin_fmaf(R[7], Q.rf, R[5]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { den[j] = __builtin_fmaf(R[6], Q.cf[j], rowden); klen[j] = R[8] * Q.len[j]; }
      E = R[9]; Esolid = R[10];
    }
  } else if (TYPE == SRH_PRIM_TRIANGLE) {
    const float r0 = __builtin_fmaf(R[1], Q.rf, R[2]);
    const float r1 = __builtin_fmaf(R[5], Q.rf, R[6]);
    const float r2 = __builtin_fmaf(R[9], Q.rf, R[10]);
    const float rowden = __builtin_fmaf(R[11], Q.rf, R[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float e0 = __builtin_fmaf(R[0], Q.cf[j], r0);
      const float e1 = __builtin_fmaf(R[4], Q.cf[j], r1);
      const float e2 = __builtin_fmaf(R[8], Q.cf[j], r2);
      cand[j] = fminf(fminf(e0, e1), e2) >= 0.0f;
      den[j] = __builtin_fmaf(R[7], Q.cf[j], rowden);
      klen[j] = R[12] * Q.len[j];
    }
    E = R[13]; Esolid = R[14];
  } else {
    const float rowden = __builtin_fmaf(R[2], Q.rf, R[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cand[j] = true;
      den[j] = __builtin_fmaf(R[1], Q.cf[j], rowden);
      klen[j] = R[3] * Q.len[j];
    }
    E = R[4]; Esolid = R[5];
  }
  if (!PRETEST || TYPE == SRH_PRIM_SPHERE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { loose[j] = cand[j]; lo[j] = 0.0f; }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float inv = __builtin_amdgcn_rcpf(den[j]);
    const float t = klen[j] * inv;
    const float eps = __builtin_fmaf(E, fabsf(inv), 9.5367431640625e-7f);            // E/|den| + 2^-20
    lo[j] = __builtin_fmaf(-t, eps, t);
    const bool solid = fabsf(den[j]) >= Esolid;
    loose[j] = cand[j] && !solid;
    cand[j] = cand[j] && solid && t > 0.0f;
  }
}

// LIST_INDEXED: `list` holds global primitive indices, records are fetched from the batch's rec32 array.
template <int TYPE, bool RESWEEP, bool PRETEST>
__device__ __forceinline__ void sweep_list(const FrameDev& F, const SegDev& S, const uint32_t* __restrict__ list,
                                           uint32_t n, QuadState& Q, const double (&d)[4][3],
                                           double (&best)[4], int (&besti)[4]) {
  const float inf = __builtin_inff();
  if (n == 0) return;
  // software pipeline: the record of entry i+1 is in flight while entry i is evaluated
  int g_next = (int)list[0];
  RejectRecord<TYPE> R_next;
  R_next.load(S.rec32 + (size_t)(g_next - S.first) * kRec32Stride[TYPE]);
  for (uint32_t i = 0; i < n; ++i) {
    const int g = g_next;
    const RejectRecord<TYPE> R = R_next;
    if (i + 1 < n) {
      g_next = (int)list[i + 1];
      R_next.load(S.rec32 + (size_t)(g_next - S.first) * kRec32Stride[TYPE]);
    }
    bool cand[4], loose[4];
    float lo[4];
    pair_bounds<TYPE, PRETEST>(R, Q, cand, loose, lo);
    const bool any_loose = loose[0] || loose[1] || loose[2] || loose[3];
    if (!RESWEEP) {
      // candidates without a usable bound are confirmed on the spot: rare, and it keeps them from posing as
      // every pixel's front candidate
      if (any_loose) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (loose[j]) confirm_global(F, g, d[j], best[j], besti[j]);
      }
      const bool any_cand = cand[0] || cand[1] || cand[2] || cand[3];
      if (__builtin_amdgcn_ballot_w64(any_cand) == 0ull) continue;       // wave-uniform skip
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = cand[j] ? lo[j] : inf;
        const bool front = x < Q.lo1[j];
        Q.lo2[j] = __builtin_amdgcn_fmed3f(Q.lo1[j], x, Q.lo2[j]);
        Q.lo1[j] = fminf(Q.lo1[j], x);
        Q.g1[j] = front ? g : Q.g1[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (Q.again[j] && cand[j] && lo[j] <= Q.bound[j] && g != Q.g1[j]) {
          confirm_global(F, g, d[j], best[j], besti[j]);
          Q.bound[j] = float_above(best[j]);
        }
      }
    }
  }
}

template <bool RESWEEP, bool PRETEST>
__device__ __forceinline__ void sweep_tile(const FrameDev& F, int tile, QuadState& Q,
                                           const double (&d)[4][3], double (&best)[4], int (&besti)[4]) {
  // per object batch: its frame-wide `large` primitives, then this tile's bin
  for (int s = 0; s < F.nseg; ++s) {
    const SegDev& S = F.seg[s];
    const int bin = s * F.ntiles_pad + tile;
    const uint32_t lo = F.tile_off[bin], hi = F.tile_off[bin + 1];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const uint32_t* list = pass == 0 ? F.large + S.first : F.entries + lo;
#ifdef SRH_ABL_NOLOOP
      const uint32_t n = 0;
#else
      const uint32_t n = pass == 0 ? F.counters[s] : hi - lo;
#endif
      switch (S.type) {
        case SRH_PRIM_DISK: sweep_list<SRH_PRIM_DISK, RESWEEP, PRETEST>(F, S, list, n, Q, d, best, besti); break;
        case SRH_PRIM_PLANE: sweep_list<SRH_PRIM_PLANE, RESWEEP, PRETEST>(F, S, list, n, Q, d, best, besti); break;
        case SRH_PRIM_SPHERE: sweep_list<SRH_PRIM_SPHERE, RESWEEP, PRETEST>(F, S, list, n, Q, d, best, besti); break;
        default: sweep_list<SRH_PRIM_TRIANGLE, RESWEEP, PRETEST>(F, S, list, n, Q, d, best, besti); break;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_render_binned(FrameDev F, float* __restrict__ image,
                                                        float* __restrict__ depth, int32_t* __restrict__ nearest) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= F.ntiles) return;               // waves are independent: no LDS, no barrier
  const int tx = tile % F.tiles_x, ty = tile / F.tiles_x;
  const int c0 = tx * kTile + 4 * (lane & 3);
  const int r_raw = F.row0 + ty * kTile + (lane >> 2);
  const int r = min(r_raw, F.row1 - 1);
  QuadState Q;
  Q.rf = (float)r;
  double d[4][3];
  double best[4];
  int besti[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = min(c0 + j, F.W - 1);
    Q.cf[j] = (float)c;
    Q.len[j] = (float)pixel_ray(F, c, r, d[j]);
    Q.lo1[j] = Q.lo2[j] = __builtin_inff();
    Q.g1[j] = -1;
    best[j] = __builtin_inf();
    besti[j] = 0x7fffffff;
  }
#ifdef SRH_ABL_NOPRETEST
  const bool pretest = false;
#else
  const bool pretest = F.near_clip > 0.0;     // with near <= 0 a negative t can be valid: confirm every candidate
#endif

  if (pretest) sweep_tile<false, true>(F, tile, Q, d, best, besti);
  else sweep_tile<false, false>(F, tile, Q, d, best, besti);

  bool any_again = false;
#ifndef SRH_ABL_NOCONFIRM
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    Q.again[j] = false;
    Q.bound[j] = float_above(best[j]);        // depth confirmed on the spot during the sweep, or +inf
    if (Q.g1[j] >= 0 && Q.lo1[j] <= Q.bound[j]) {
      confirm_global(F, Q.g1[j], d[j], best[j], besti[j]);
      Q.bound[j] = float_above(best[j]);
      Q.again[j] = Q.lo2[j] <= Q.bound[j];    // somebody else's bound still reaches the confirmed depth
    }
    any_again = any_again || Q.again[j];
  }
#endif
#ifndef SRH_ABL_NORESWEEP
  if (__builtin_amdgcn_ballot_w64(any_again) != 0ull) sweep_tile<true, true>(F, tile, Q, d, best, besti);
#endif

  const bool row_live = r_raw < F.row1;
  const size_t row = (size_t)(r - F.row0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (besti[j] == 0x7fffffff) besti[j] = 0;             // nothing hit: np.argmin of an all-inf column
    float rgb[3];
#ifdef SRH_ABL_NOSHADE
    rgb[0] = rgb[1] = rgb[2] = (float)d[j][0] + Q.lo1[j];
#else
    shade_pixel(F, d[j], best[j], besti[j], rgb);
#endif
    const int c = c0 + j;
    if (row_live && c < F.W) {
      float* px = image + row * F.img_stride + 3 * (size_t)c;
      px[0] = rgb[0]; px[1] = rgb[1]; px[2] = rgb[2];
      depth[row * F.depth_stride + c] = (float)best[j];
#ifdef SRH_DIAG_G1      // diagnostic build: the front candidate and its depth lower bound
      if (nearest) nearest[row * F.near_stride + c] = Q.g1[j];
      depth[row * F.depth_stride + c] = Q.lo1[j];
#elif defined(SRH_DIAG_AGAIN)   // diagnostic build: why did this pixel ask for a re-sweep (0 = it did not)
      if (nearest) nearest[row * F.near_stride + c] = !Q.again[j] ? 0 : (Q.bound[j] == __builtin_inff() ? 1 : 2);
#else
      if (nearest) nearest[row * F.near_stride + c] = besti[j];
#endif
    }
  }
}

}  // namespace srh

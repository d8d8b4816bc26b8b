// BINNED mode kernels (gfx950): screen-space tile binning + one workgroup per 16x16-pixel tile.
//
//   k_prep (srh.hip)   also writes each primitive's tile range and either counts it into the tiles it
//                      overlaps (atomicAdd on per-tile counters) or appends it to the `large` list
//   k_bin_scan         exclusive prefix sum of the per-tile counts (one workgroup)
//   k_bin_fill         second pass over the primitives: claim a slot per overlapped tile, write the index
//   k_render_binned    per tile: stage the tile's primitives (and, first, the `large` ones) through LDS in
//                      batches of 256 reject records, every thread owns one pixel and keeps its fp64 ray in
//                      registers; fp32 reject per pair, fp64 confirmation of survivors, shade, store
//
// Bin contents come out of atomics in arbitrary order; the winner is the lexicographic minimum of
// (t, global index), which does not depend on visiting order, so frames are bit-reproducible and
// identical to the ordered all-pairs modes.
#pragma once
#include "srh_device.h"
#include "srh_reject.h"

namespace srh {

// ---- tile range of one primitive (called from k_prep) ------------------------------------------------
__device__ inline void bin_primitive(const FrameDev& F, int type, const float* rec32, int gidx) {
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
  if (is_large) {
    const uint32_t slot = atomicAdd(&F.counters[0], 1u);
    F.large[slot] = (uint32_t)gidx;
    return;
  }
  tr[0] = (uint16_t)tx0; tr[1] = (uint16_t)ty0; tr[2] = (uint16_t)tx1; tr[3] = (uint16_t)ty1;
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&F.counters[kCounterPad + ty * F.tiles_x + tx], 1u);
}

// ---- exclusive scan of the tile counts: one 1024-thread workgroup ---------------------------------------
__global__ __launch_bounds__(1024) void k_bin_scan(FrameDev F) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  const int n = F.ntiles;
  const int chunk = (n + 1023) / 1024;
  const int lo = tid * chunk, hi = min(lo + chunk, n);
  const uint32_t* cnt = F.counters + kCounterPad;
  uint32_t sum = 0;
  for (int i = lo; i < hi; ++i) sum += cnt[i];
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {             // Hillis-Steele inclusive scan of the partials
    const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;                         // exclusive prefix of this thread's chunk
  for (int i = lo; i < hi; ++i) {
    F.tile_off[i] = run;
    run += cnt[i];
  }
  if (tid == 1023) F.tile_off[n] = part[1023];
}

// ---- fill: one thread per primitive ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bin_fill(FrameDev F) {
  const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= F.total) return;
  const uint16_t* tr = F.tilerange + 4 * (size_t)gidx;
  const int tx0 = tr[0], ty0 = tr[1], tx1 = tr[2], ty1 = tr[3];
  if (tx0 > tx1) return;
  uint32_t* cursor = F.counters + kCounterPad + F.ntiles;
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) {
      const int t = ty * F.tiles_x + tx;
      const uint32_t slot = atomicAdd(&cursor[t], 1u);
      F.entries[F.tile_off[t] + slot] = (uint32_t)gidx;
    }
}

// ---- render: one workgroup per tile -------------------------------------------------------------------------
struct alignas(16) StagedPrim {   // 64 bytes of LDS per staged primitive
  float q[12];                    // reject record (layouts in srh_reject.h)
  int32_t gidx, seg, local, type;
};

__device__ __forceinline__ void stage_primitive(const FrameDev& F, uint32_t gidx, StagedPrim* dst) {
  int seg = 0, type = 0, local = 0;
  const float* src = nullptr;
#pragma unroll
  for (int s = 0; s < SRH_MAX_SEGMENTS; ++s) {
    if (s < F.nseg && (int)gidx >= F.seg[s].first && (int)gidx < F.seg[s].first + F.seg[s].count) {
      seg = s;
      type = F.seg[s].type;
      local = (int)gidx - F.seg[s].first;
      src = F.seg[s].rec32 + (size_t)local * kRec32Stride[type];
    }
  }
  const float4* src4 = reinterpret_cast<const float4*>(src);
  float4* dst4 = reinterpret_cast<float4*>(dst->q);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  dst4[0] = src4[0];
  dst4[1] = (type != SRH_PRIM_PLANE) ? src4[1] : zero;
  dst4[2] = (type == SRH_PRIM_TRIANGLE) ? src4[2] : zero;
  dst->gidx = (int)gidx;
  dst->seg = seg;
  dst->local = local;
  dst->type = type;
}

__device__ __forceinline__ void test_staged(const FrameDev& F, const StagedPrim& R, float cf, float rf,
                                            const double d[3], double& best, int& besti) {
  const int type = __builtin_amdgcn_readfirstlane(R.type);
  bool cand = true;
  if (type == SRH_PRIM_DISK || type == SRH_PRIM_SPHERE) {
    const float dc = cf - R.q[0], dr = rf - R.q[1];
    const float e = R.q[3] * dr;
    const float g = __builtin_fmaf(R.q[4] * dr, dr, -1.0f);
    cand = __builtin_fmaf(dc, __builtin_fmaf(R.q[2], dc, e), g) <= 0.0f;
  } else if (type == SRH_PRIM_TRIANGLE) {
    const float e0 = __builtin_fmaf(R.q[0], cf, __builtin_fmaf(R.q[1], rf, R.q[2]));
    const float e1 = __builtin_fmaf(R.q[4], cf, __builtin_fmaf(R.q[5], rf, R.q[6]));
    const float e2 = __builtin_fmaf(R.q[8], cf, __builtin_fmaf(R.q[9], rf, R.q[10]));
    cand = fminf(fminf(e0, e1), e2) >= 0.0f;
  }
  if (cand) {
    const int seg = __builtin_amdgcn_readfirstlane(R.seg);
    const int local = __builtin_amdgcn_readfirstlane(R.local);
    const int gidx = __builtin_amdgcn_readfirstlane(R.gidx);
    const double* rec = F.seg[seg].rec64 + (size_t)local * kRec64Stride[type];
    resolve_lex(F, hit_any64(type, rec, F.o, d), gidx, best, besti);
  }
}

__global__ __launch_bounds__(256) void k_render_binned(FrameDev F, float* __restrict__ image,
                                                        float* __restrict__ depth, int32_t* __restrict__ nearest) {
  __shared__ StagedPrim staged[256];
  const int tid = threadIdx.x;
  const int tile = blockIdx.y * F.tiles_x + blockIdx.x;
  const int c_raw = blockIdx.x * kTile + (tid & (kTile - 1));
  const int r_raw = F.row0 + blockIdx.y * kTile + (tid >> 4);
  const bool live = (c_raw < F.W) && (r_raw < F.row1);
  const int c = min(c_raw, F.W - 1), r = min(r_raw, F.row1 - 1);
  const float cf = (float)c, rf = (float)r;
  double d[3];
  pixel_ray(F, c, r, d);
  double best = __builtin_inf();
  int besti = 0x7fffffff;

  // two lists per tile: the frame-wide `large` primitives, then this tile's bin
  const uint32_t n_large = F.counters[0];
  const uint32_t bin_lo = F.tile_off[tile], bin_hi = F.tile_off[tile + 1];
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const uint32_t* list = pass == 0 ? F.large : F.entries + bin_lo;
    const uint32_t n = pass == 0 ? n_large : bin_hi - bin_lo;
    for (uint32_t base = 0; base < n; base += 256) {
      const uint32_t m = min(256u, n - base);
      __syncthreads();                       // everyone is done with the previous batch
      if ((uint32_t)tid < m) stage_primitive(F, list[base + tid], &staged[tid]);
      __syncthreads();
      for (uint32_t e = 0; e < m; ++e) test_staged(F, staged[e], cf, rf, d, best, besti);
    }
  }
  if (besti == 0x7fffffff) besti = 0;        // nothing hit: np.argmin of an all-inf column
  float rgb[3];
  shade_pixel(F, d, best, besti, rgb);
  if (live) {
    const size_t row = (size_t)(r_raw - F.row0);
    float* px = image + row * F.img_stride + 3 * (size_t)c_raw;
    px[0] = rgb[0]; px[1] = rgb[1]; px[2] = rgb[2];
    depth[row * F.depth_stride + c_raw] = (float)best;
    if (nearest) nearest[row * F.near_stride + c_raw] = besti;
  }
}

}  // namespace srh

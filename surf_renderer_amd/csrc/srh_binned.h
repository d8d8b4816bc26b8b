// BINNED mode kernels (gfx950): screen-space tile binning + one WAVE per 16x16-pixel tile.
//
//   k_prep (srh.hip)   builds a primitive's records, finds its box of tiles (or appends it to its batch's `large`
//                      list) and PLACES it: for every tile of the box the reject shape really reaches, an atomicAdd on
//                      the bin's counter claims a slot of the bin's fixed-capacity list and the index goes there
//                      (bin_place).  One pass over the primitives, no count / scan / fill.  A bin that is full sends
//                      the primitive to the `large` list instead, which every tile tests.
//   k_bin_count        the placement as a kernel of its own, two lanes per primitive (build switch SRH_FUSE_BIN=0)
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
struct TileBox { int tx0, ty0, tx1, ty1; };        // inclusive; tx0 > tx1: not binned (no pixel, or on the large list)

__device__ inline TileBox bin_primitive(const FrameDev& F, int seg, int type, const float* rec32, int gidx,
                                        uint32_t set, bool force_large = false) {
  BBox b = bbox_full();
  if (force_large) { /* keep the full box */ }
  else if (type == SRH_PRIM_DISK || type == SRH_PRIM_SPHERE) b = conic_bbox(rec32);
  else if (type == SRH_PRIM_TRIANGLE) b = triangle_bbox(rec32);
  uint16_t* tr = F.tilerange + 4 * (size_t)gidx;
  tr[0] = 1; tr[1] = 0; tr[2] = 0; tr[3] = 0;                    // default: not binned
  bool is_large = b.full;
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  if (!b.full) {
    // inclusive pixel box, clamped to the rendered slab; empty -> the primitive touches no pixel at all
    const double c_lo = fmax(floor(b.c0), 0.0), c_hi = fmin(ceil(b.c1), (double)(F.W - 1));
    const double r_lo = fmax(floor(b.r0), (double)F.row0), r_hi = fmin(ceil(b.r1), (double)(F.row1 - 1));
    if (!(c_lo <= c_hi) || !(r_lo <= r_hi)) return TileBox{1, 0, 0, 0};
    tx0 = (int)c_lo / kTile; tx1 = (int)c_hi / kTile;
    ty0 = ((int)r_lo - F.row0) / kTile; ty1 = ((int)r_hi - F.row0) / kTile;
    is_large = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) > kMaxTilesPerPrim;
  }
  if (is_large) {                                                 // the batch's region of `large` starts at seg.first
    // bounded like the bin lists: whatever the counter holds (a workspace whose counters were never cleared), the
    // store stays inside the batch's region -- every primitive joins at most once, so a clean counter never exceeds it
    const uint32_t slot = atomicAdd(&F.counters[4u * set + seg], 1u);
    if (slot < (uint32_t)F.seg[seg].count) F.large[F.seg[seg].first + slot] = (uint32_t)gidx;
    return TileBox{1, 0, 0, 0};
  }
  tr[0] = (uint16_t)tx0; tr[1] = (uint16_t)ty0; tr[2] = (uint16_t)tx1; tr[3] = (uint16_t)ty1;
  return TileBox{tx0, ty0, tx1, ty1};
}

__device__ __forceinline__ int rec32_stride(int type) {
  return type == SRH_PRIM_DISK ? kRec32Stride[0] : type == SRH_PRIM_PLANE ? kRec32Stride[1]
       : type == SRH_PRIM_SPHERE ? kRec32Stride[2] : kRec32Stride[3];
}

// Workgroup size of the per-primitive kernels (prep, count): ONE wave.  They run beside other frames' render
// kernels, whose single-wave workgroups refill every slot the moment it is free; a four-wave workgroup would wait for
// four free slots on one CU and starve.  (Measured together with SRH_GROUP_WAVES below: 256/4 0.0850 ms per frame at
// config 5, 256/1 0.0948-0.0980, 64/1 0.0816.)
#ifndef SRH_BIN_BLOCK
#define SRH_BIN_BLOCK 64
#endif
constexpr int kBinBlock = SRH_BIN_BLOCK;
// Place the primitive from k_prep itself (a thread per primitive walks its box right after it built the record): no
// count kernel, and nobody reads the reject record back from another XCD.
#ifndef SRH_FUSE_BIN
#define SRH_FUSE_BIN 1
#endif
#ifndef SRH_COUNT_LANES
#define SRH_COUNT_LANES 2
#endif
constexpr int kCountLanes = SRH_COUNT_LANES;

// ---- count: kCountLanes lanes per primitive, one tile of its box per lane and step ------------------------------
// Of the (at most 64) tiles of the box keep those the reject shape really reaches and that are not provably behind
// the eye; bit k of the primitive's tile mask = k-th tile of the box, row-major.  k_prep leaves this to a kernel of
// its own because a thread per primitive walking up to 64 tiles in fp64 is a long serial chain on a launch that
// has only a wave or two per SIMD.
// The tiles of primitive gidx's box that its reject shape really reaches: kLanes lanes share the box (lane `sub` takes
// tiles sub, sub + kLanes, ...).
#ifndef SRH_BIN_CLAIMS
#define SRH_BIN_CLAIMS 4
#endif
constexpr int kClaims = SRH_BIN_CLAIMS;
template <int kLanes>
__device__ __forceinline__ void bin_place(const FrameDev& F, int seg, int type, int first, const float* rec32, int gidx,
                                          int sub, int tx0, int ty0, int tx1, int ty1, uint32_t set) {
  uint32_t* count = F.counters + kCounterPad + seg * F.ntiles_pad;
  const int nx = tx1 - tx0 + 1, n = nx * (ty1 - ty0 + 1);
  const RectTest T(type, rec32, F.near_clip > 0.0);
  // pass 1, arithmetic only: bit k of `mask` = tile k of the box (row-major) is reached
  const uint32_t inv_nx = 65536u / (uint32_t)nx + 1u;            // k / nx == (k * inv_nx) >> 16 for k < 64, nx <= 64
  uint64_t mask = 0;
  for (int k = sub; k < n; k += kLanes) {
    const int dy = (int)(((uint32_t)k * inv_nx) >> 16);
    const int tx = tx0 + (k - dy * nx), ty = ty0 + dy;
    const double pc0 = tx * kTile - F.bin_pad, pr0 = F.row0 + ty * kTile - F.bin_pad;
    const double pc1 = fmin(tx * kTile + kTile - 1, (double)(F.W - 1)) + F.bin_pad;
    const double pr1 = fmin(F.row0 + ty * kTile + kTile - 1, (double)(F.row1 - 1)) + F.bin_pad;
    if (T.reaches(pc0, pc1, pr0, pr1)) mask |= 1ull << k;
  }
  // pass 2: the slot claimed in a bin's counter is the entry's place in the bin's fixed-capacity list.  kClaims claims
  // at a time, so that their round trips overlap instead of adding up (a thread owns ~6 bins on average).
  const uint32_t cap = (uint32_t)F.bin_cap;
  uint32_t* slots = F.entries + (size_t)seg * F.ntiles_pad * cap;
  int over = 0;
  while (mask) {
    uint32_t tile[kClaims], slot[kClaims];
#pragma unroll
    for (int j = 0; j < kClaims; ++j) {
      tile[j] = 0xffffffffu;
      if (mask) {
        const uint32_t k = (uint32_t)__builtin_ctzll(mask);
        mask &= mask - 1;
        const uint32_t dy = (k * inv_nx) >> 16, dx = k - dy * (uint32_t)nx;
        tile[j] = (uint32_t)(ty0 + (int)dy) * (uint32_t)F.tiles_x + (uint32_t)(tx0 + (int)dx);
        slot[j] = atomicAdd(&count[tile[j]], 1u);
      }
    }
#pragma unroll
    for (int j = 0; j < kClaims; ++j)
      if (tile[j] != 0xffffffffu) {
        if (slot[j] < cap) slots[(size_t)tile[j] * cap + slot[j]] = (uint32_t)gidx;
        else over = 1;
      }
  }
  // a full bin: the primitive joins the batch's `large` list (tested by every tile) -- once, whatever the number of
  // full bins; the bins that did take it keep it, a duplicate candidate changes no result
#pragma unroll
  for (int m = 1; m < kLanes; m <<= 1) over |= __shfl_xor(over, m);
  if (sub == 0 && over) {
    const uint32_t at = atomicAdd(&F.counters[4u * set + seg], 1u);
    if (at < (uint32_t)F.seg[seg].count) F.large[first + at] = (uint32_t)gidx;
  }
}

__device__ __forceinline__ void bin_count_body(const FrameDev& F) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int gidx = t / kCountLanes, sub = t % kCountLanes;
  if (gidx >= F.total) return;                                  // whole lane groups leave together
  const uint16_t* tr = F.tilerange + 4 * (size_t)gidx;
  const int tx0 = tr[0], ty0 = tr[1], tx1 = tr[2], ty1 = tr[3];
  if (tx0 > tx1) return;
  const int seg = segment_of(F, gidx);
  int type = F.seg[0].type, first = F.seg[0].first;
  const float* base = F.seg[0].rec32;
#pragma unroll
  for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
    if (seg == i) { type = F.seg[i].type; first = F.seg[i].first; base = F.seg[i].rec32; }
  const float* rec32 = base + (size_t)(gidx - first) * rec32_stride(type);
  bin_place<kCountLanes>(F, seg, type, first, rec32, gidx, sub, tx0, ty0, tx1, ty1, F.counters[kLargeNext] & 1u);
}

// the bin's list and its length (bin = seg * ntiles_pad + tile)
__device__ __forceinline__ const uint32_t* bin_list(const FrameDev& F, int bin) {
  return F.entries + (size_t)bin * (uint32_t)F.bin_cap;
}
__device__ __forceinline__ uint32_t bin_length(const FrameDev& F, int bin) {
  return min(F.counters[kCounterPad + bin], (uint32_t)F.bin_cap);
}

// length of batch s's frame-wide list, clamped to the batch like bin_length to the bin: reads stay inside the workspace
__device__ __forceinline__ uint32_t large_length(const FrameDev& F, int s) {
  return min(F.counters[4u * (F.counters[kLargeNow] & 1u) + s], (uint32_t)F.seg[s].count);
}

__global__ __launch_bounds__(kBinBlock) void k_bin_count(FrameDev F) { bin_count_body(F); }

// counters <- 0 (four words per thread)
__global__ __launch_bounds__(256) void k_zero_counters(uint32_t* __restrict__ counters, uint32_t n) {
  const uint32_t i = 4u * (blockIdx.x * blockDim.x + threadIdx.x);
  if (i + 3 < n) *reinterpret_cast<uint4*>(counters + i) = make_uint4(0u, 0u, 0u, 0u);
  else
    for (uint32_t k = i; k < n; ++k) counters[k] = 0u;
}

// ---- render: one wave per 16x16-pixel tile, four pixels (one row quad) per lane ---------------------------------
//
// sweep     wave-uniform loop over the tile's lists, fp32 only: per (pixel, primitive) pair the screen-space
//           reject and, for the survivors, a LOWER BOUND of the ray distance (plane_estimate_record).  Per pixel
//           the four smallest bounds are kept (packed keys, see QuadState); a candidate without a usable
//           bound counts as bound 0.  List words and reject records are wave-uniform: they arrive by
//           scalar loads, software-pipelined one record and two list words ahead.  No fp64 in this loop.
// finish    per pixel (state parked in LDS so only one pixel's fp64 state is live): the front candidate goes
//           through the fp64 intersection; the second and third too while their bound still reaches the depth
//           confirmed so far (a near miss at an ellipse edge, or nearly coplanar primitives).  If the fourth
//           bound lies above the confirmed depth nobody else can win or tie.  Then shade and store.
// slow path (very rare) for a pixel whose fourth bound reaches the confirmed depth the whole wave walks the tile's lists
//           again, 64 entries at a time, confirms every other candidate that is unbounded or whose bound reaches
//           the confirmed depth, and merges the results with a wavefront-shuffle lexicographic minimum.
// A candidate is skipped only when its lower bound exceeds an exactly confirmed depth, so it can neither win nor
// tie: the output equals the all-pairs fp64 mode bit for bit.
// Per pixel the sweep keeps the four LARGEST keys.  A key packs an UPPER bound of a candidate's inverse ray distance
// and its position in the tile's lists into one 32-bit word, so that tracking is four integer max/median operations
// and the sweep needs no reciprocal:
//   inv = den * rlen                         den: affine estimate (plane_estimate_record), rlen = 1 / |D|;
//                                            inv >= 1 / t for a valid hit, inv <= 0 proves there is none (near > 0)
//   key = (bits(inv) & ~0xFFF) | field       compared as SIGNED integers: positive floats order like their bit
//                                            patterns, negative ones are negative and never displace the initial 0.
//                                            Decoded with the low bits SET, so it is still an upper bound of 1 / t
//   key = (bits(1e30) & ~0xFFF) | field      candidate without an estimate (sphere, near <= 0, withdrawn): ranks first
//   key = 0                                  not a candidate
// field = 1 + index of the primitive in the concatenation of the tile's lists, saturated at 4095 (a pixel whose
// front keys carry the saturated field takes the slow path).
// Cost model behind the arithmetic (measured on MI355X, ns per wave instruction and SIMD): v_add/sub/mul_f32,
// and/or/xor/not, ashr and integer add on VGPRs 1.0; every VOP3, fma, min/max/med3, compares, bfi and the packed
// f32 forms 1.8 (a packed fma does two pixels); v_rcp_f32 3.5; cmp + cndmask 2.9 against ashr + and 2.1.
constexpr uint32_t kOrdMask = 0xFFFu;
constexpr int32_t kNoKey = 0;
constexpr float kNoEstimate = 1.0e30f;
#ifndef SRH_SERIAL_SLOW_MAX
#define SRH_SERIAL_SLOW_MAX 2
#endif
constexpr int kSerialSlowMax = SRH_SERIAL_SLOW_MAX;   // more undecided pixels than this in a round: lane-parallel re-sweep
#ifndef SRH_KEYS
#define SRH_KEYS 4
#endif
constexpr int kKeys = SRH_KEYS;             // keys tracked per pixel: kKeys - 1 are confirmed in fp64, the last is a bound only

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct QuadState {
  uint32_t ordmask;   // wave-uniform: the low key bits that hold the list position (ord_mask_for), <= kOrdMask
  f32x2 cf[2];        // pixel columns as fp32 (exact integers): (c, c+1), (c+2, c+3)
  float rf;           // pixel row
  f32x2 rlen[2];      // 1 / |D| of the un-normalised ray direction
  int32_t k1[4], k2[4], k3[4], k4[4];   // four largest keys, descending
};

// median of three.  Spelled with min/max hipcc shares a min with the neighbouring key update and ends up with four
// min/max per key instead of one v_med3_i32, so the instruction is named.
__device__ __forceinline__ int32_t imed3(int32_t a, int32_t b, int32_t c) {
  int32_t r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// The key's position field is as wide as THIS tile's lists need, not a fixed 12 bits: a tile with n entries uses
// b = bits(n + 1) low bits (mask 2^b - 1 >= n + 1, so no field saturates), at most kOrdMask's 12.  Every bit the field does
// not take stays with the inverse-depth bound: at BASELINE config 5 (41 entries per busy tile on average, 103 at most)
// the bound keeps 16-17 mantissa bits instead of 11, and two candidates have to be 32-64 times closer in depth before
// the finish has to confirm both.  With 12 fixed bits the second confirmation ran in 4 of 10 finish rounds (one
// ambiguous pixel among a round's 64 is enough); it is ~60 fp64 instructions for the whole wave.
__device__ __forceinline__ uint32_t ord_mask_for(uint32_t n_entries) {
  const uint32_t bits = 32u - (uint32_t)__builtin_clz(n_entries + 1u);        // n_entries + 1 >= 1
  return min((1u << bits) - 1u, kOrdMask);
}

// (bits(inv) & ~mask) | field in one v_bfi_b32 (field <= mask, in a vector register; `keep` = ~mask, wave-uniform)
__device__ __forceinline__ int32_t pack_key(float inv, uint32_t field, uint32_t keep) {
  int32_t r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(keep), "v"(inv), "v"(field));
  return r;
}

__device__ __forceinline__ bool key_saturated(int32_t key, uint32_t mask) { return ((uint32_t)key & mask) == mask; }
__device__ __forceinline__ uint32_t key_ordinal(int32_t key, uint32_t mask) { return ((uint32_t)key & mask) - 1u; }

// fp32 LOWER bound of the ray distance a key stands for (rcp is good to 1 ulp; the factor covers it)
// Does a candidate whose inverse-depth bound is `inv` (>= 1 / t of any valid hit of it) still "reach" a pixel whose
// confirmed depth is at most `bound` -- could it win or tie?  Only if inv >= 1 / bound; compared against
//   reach = min((1 - 1e-6) / bound, kReachMax).
// The clamp makes candidates WITHOUT an estimate (inv = kNoEstimate, or a withdrawn estimate's 1e30 / |D|) reach
// every depth, including bound = 0 and negative depths, which are valid hits when near <= 0 (the numpy backend's
// sphere sentinel t = 0 ties across spheres and must go to the lowest index); a real estimate stays below kReachMax
// unless the hit is closer than 1e-25, where it reaches anyway.  bound = +inf (nothing confirmed) gives reach 0.
constexpr float kReachMax = 1.0e25f;
__device__ __forceinline__ float reach_of(float bound) { return fminf(__builtin_amdgcn_rcpf(bound) * 0.999999f, kReachMax); }
__device__ __forceinline__ float key_inv(int32_t key, uint32_t mask) { return __uint_as_float((uint32_t)key | mask); }   // >= 1 / t

// fp32 value that is certainly >= the fp64 depth (the conversion may round down by half an ulp)
__device__ __forceinline__ float float_above(double t) { return (float)t * 1.0000005f; }

// ONE (here and below): the kernel instantiation for scenes with one object batch -- per-batch loops and per-lane batch
// selections fold away at compile time
template <bool TCH, int BATCH = -1>
__device__ __forceinline__ void confirm_global(const FrameDev& F, int gidx, const double d[3], double& best,
                                               int& besti) {
  if ((BATCH >= 0) || F.nseg == 1) {
    // one object batch (the common scene): type and base are wave-uniform, the intersection is chosen by a scalar branch
    const int type = BATCH >= 0 ? BATCH : F.seg[0].type;
    const double* R = F.seg[0].rec64 + (size_t)(gidx - F.seg[0].first) * kRec64Stride[type];
    resolve_lex(F, hit_any64(type, R, F.o, d, TCH), gidx, best, besti);
    return;
  }
  const int s = segment_of(F, gidx);
  int type = F.seg[0].type, first = F.seg[0].first;
  const double* base = F.seg[0].rec64;
#pragma unroll
  for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
    if (s == i) { type = F.seg[i].type; first = F.seg[i].first; base = F.seg[i].rec64; }
  const double* R = base + (size_t)(gidx - first) * kRec64Stride[type];
  resolve_lex(F, hit_any64(type, R, F.o, d, TCH), gidx, best, besti);
}

// The fp64 record of a pixel's front candidate, fetched ahead of its use (see the finish rounds): up to eight doubles
// in registers (all of a disc, plane or sphere record; normal and plane offset of a triangle), plus the material
// index for the fragment stage.  g < 0 (no front candidate) fetches primitive 0, and nobody looks at the result.
template <bool TCH, int BATCH = -1>
struct FrontRecord {
  int g, type, m;
  const double* R;
  double v[8];
  __device__ __forceinline__ void fetch(const FrameDev& F, int gidx) {
    g = gidx;
    const int gs = max(gidx, 0);
    if ((BATCH >= 0) || F.nseg == 1) {               // one batch: type, stride and bases are wave-uniform
      const int t0 = BATCH >= 0 ? BATCH : F.seg[0].type;
      type = t0;
      const int li = gs - F.seg[0].first;
      R = F.seg[0].rec64 + (size_t)li * kRec64Stride[t0];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = R[i];
      if (t0 == SRH_PRIM_DISK || t0 == SRH_PRIM_TRIANGLE) {
#pragma unroll
        for (int i = 4; i < 8; ++i) v[i] = R[i];
      } else {
#pragma unroll
        for (int i = 4; i < 8; ++i) v[i] = 0.0;
      }
      m = clampi(F.seg[0].mat[li], 0, F.nmat - 1);
      return;
    }
    int first = F.seg[0].first;
    const double* base = F.seg[0].rec64;
    const int32_t* mat = F.seg[0].mat;
    type = F.seg[0].type;
    if (!(BATCH >= 0) && F.nseg > 1) {               // one batch: everything above is already right (and wave-uniform)
      const int s = segment_of(F, gs);
#pragma unroll
      for (int i = 1; i < SRH_MAX_SEGMENTS; ++i)
        if (s == i) { type = F.seg[i].type; first = F.seg[i].first; base = F.seg[i].rec64; mat = F.seg[i].mat; }
    }
    const int li = gs - first;
    const int stride = type == SRH_PRIM_DISK ? kRec64Stride[0] : type == SRH_PRIM_PLANE ? kRec64Stride[1]
                     : type == SRH_PRIM_SPHERE ? kRec64Stride[2] : kRec64Stride[3];
    R = base + (size_t)li * stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = R[i];
    if (type == SRH_PRIM_DISK || type == SRH_PRIM_TRIANGLE) {
#pragma unroll
      for (int i = 4; i < 8; ++i) v[i] = R[i];
    } else {
#pragma unroll
      for (int i = 4; i < 8; ++i) v[i] = 0.0;
    }
    m = clampi(mat[li], 0, F.nmat - 1);
  }
  __device__ __forceinline__ double hit(const FrameDev& F, const double d[3]) const {
    if ((BATCH >= 0) || F.nseg == 1) {               // wave-uniform type: a scalar branch picks the intersection
      switch (BATCH >= 0 ? BATCH : F.seg[0].type) {
        case SRH_PRIM_DISK: return hit_disk64(v, F.o, d);
        case SRH_PRIM_PLANE: return hit_plane64(v, d);
        case SRH_PRIM_SPHERE: return TCH ? hit_sphere64_tch(v, d) : hit_sphere64(v, d);
        default: return hit_triangle64(R, F.o, d);
      }
    }
    switch (type) {
      case SRH_PRIM_DISK: return hit_disk64(v, F.o, d);
      case SRH_PRIM_PLANE: return hit_plane64(v, d);
      case SRH_PRIM_SPHERE: return TCH ? hit_sphere64_tch(v, d) : hit_sphere64(v, d);
      default: return hit_triangle64(R, F.o, d);
    }
  }
  __device__ __forceinline__ ShadeHint hint() const { return ShadeHint{g, m, {v[0], v[1], v[2]}}; }
};

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

// The lane's four pixels (one row: columns cf, row rf, inverse direction lengths rlen) against one reject record:
//   sel[j]   all ones if pixel j passes the screen-space reject, else 0
//   inv[j]   upper bound of 1 / t of a valid hit (kNoEstimate: no usable bound; <= 0: provably no valid hit)
// A pair that does not pass is provably not a valid hit.  Written on two-pixel vectors so that hipcc emits the
// packed f32 forms; the selects are sign masks (v_ashrrev_i32) instead of compare + select.
__device__ __forceinline__ f32x2 splat2(float x) { return f32x2{x, x}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// SCALE = false leaves out the factor 1 / |D|: the value is then den = inv |D|, which orders the candidates of ONE pixel
// exactly as inv does -- the typed kernels of planar primitives track their keys in that unit (kDenKeys) and convert at
// the few places that compare a key with a depth.
template <int TYPE, bool PRETEST, bool SCALE = true>
__device__ __forceinline__ void pair_bounds(const RejectRecord<TYPE>& R, const f32x2 (&cf)[2], float rf,
                                            const f32x2 (&rlen)[2], int32_t (&sel)[4], f32x2 (&inv)[2]) {
  f32x2 den[2];
  if (TYPE == SRH_PRIM_DISK || TYPE == SRH_PRIM_SPHERE) {
    // candidate iff q <= 0; tested as q - 2^-22 < 0 (a superset) so that the sign bit decides
    const float dr = rf - R[1];
    f32x2 q[2];
    // one form for every ellipse (conic_record; until round 3 elongated ones came in a principal-axes form, chosen
    // per entry by a wave-uniform branch)
    {
      const float ee = R[3] * dr;
      const float gg = __builtin_fmaf(R[4] * dr, dr, -1.00000024f);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const f32x2 dc = cf[p] - splat2(R[0]);
        q[p] = fma2(dc, fma2(splat2(R[2]), dc, splat2(ee)), splat2(gg));
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sel[j] = __float_as_int(q[j >> 1][j & 1]) >> 31;
    if (TYPE == SRH_PRIM_DISK) {
      // a multiply and an add (2.5 cycles each, one scalar operand each) instead of one fma, which would need a v_mov
      // for its second scalar operand (one constant-bus read per instruction) and costs 4.7 cycles itself
      const float rowden = R[7] * rf + R[5];
#pragma unroll
      for (int p = 0; p < 2; ++p) den[p] = fma2(splat2(R[6]), cf[p], splat2(rowden));
    }
  } else if (TYPE == SRH_PRIM_TRIANGLE) {
    // candidate iff all three edge functions are >= 0; a true hit has them > 0 (the margin in g), so the
    // largest NEGATED one is < 0 and its sign bit decides
    const float r0 = __builtin_fmaf(R[1], rf, R[2]);
    const float r1 = __builtin_fmaf(R[5], rf, R[6]);
    const float r2 = __builtin_fmaf(R[9], rf, R[10]);
    const float rowden = R[11] * rf + R[3];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x2 e0 = fma2(splat2(-R[0]), cf[p], splat2(-r0));
      const f32x2 e1 = fma2(splat2(-R[4]), cf[p], splat2(-r1));
      const f32x2 e2 = fma2(splat2(-R[8]), cf[p], splat2(-r2));
#pragma unroll
      for (int h = 0; h < 2; ++h) sel[2 * p + h] = __float_as_int(fmaxf(fmaxf(e0[h], e1[h]), e2[h])) >> 31;
      den[p] = fma2(splat2(R[7]), cf[p], splat2(rowden));
    }
  } else {
    const float rowden = R[2] * rf + R[0];
#pragma unroll
    for (int p = 0; p < 2; ++p) den[p] = fma2(splat2(R[1]), cf[p], splat2(rowden));
#pragma unroll
    for (int j = 0; j < 4; ++j) sel[j] = -1;                // every pixel; behind the eye inv <= 0 drops out by itself
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (!PRETEST) inv[p] = splat2(kNoEstimate);
    else if (TYPE == SRH_PRIM_SPHERE) inv[p] = splat2(R[5]);    // per-sphere constant (sphere_reject_record)
    else if (SCALE) inv[p] = den[p] * rlen[p];
    else inv[p] = den[p];
  }
}

// One staged primitive against the lane's four pixels: update the keys.
template <int TYPE, bool PRETEST, bool DENKEYS>
__device__ __forceinline__ void sweep_entry(const RejectRecord<TYPE>& R, uint32_t field, QuadState& Q) {
  int32_t sel[4];
  f32x2 inv[2];
  pair_bounds<TYPE, PRETEST, !DENKEYS>(R, Q.cf, Q.rf, Q.rlen, sel, inv);
  const uint32_t keep = ~Q.ordmask;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int32_t key = pack_key(inv[j >> 1][j & 1], field, keep) & sel[j];
    if (kKeys == 4) Q.k4[j] = imed3(Q.k3[j], key, Q.k4[j]);
    Q.k3[j] = imed3(Q.k2[j], key, Q.k3[j]);
    Q.k2[j] = imed3(Q.k1[j], key, Q.k2[j]);
    Q.k1[j] = max(Q.k1[j], key);
  }
}

// The hot loop: fp32 only, straight-line, updates the per-pixel keys.  `ord0` = ordinal of the list's first entry
// (the key field is the ordinal + 1).
// Scalar loads complete out of order, so the only wait is "all of them": the loop is unrolled by two over two
// record buffers A and B -- while A is evaluated, B's record (and the list word after it) is in flight, and the
// wait for B comes only after A's ~100 vector instructions.  No register copies between the buffers.
template <int TYPE, int WPT, class Op>
__device__ __forceinline__ void stream_list(const SegDev& S, const uint32_t* __restrict__ list, uint32_t n_all,
                                            uint32_t ord0, uint32_t part, Op&& op, uint32_t ordmask = kOrdMask) {
  // with WPT waves per tile this wave takes entries part, part + WPT, ...;  op(record, global index, key field)
  if (n_all <= part) return;
  const uint32_t n = (n_all - part + WPT - 1) / WPT;
  RejectRecord<TYPE> A, B;
  const int first = S.first;
  // 32-bit byte offsets, so that hipcc addresses list words and records as base + scalar offset (`s_load ... sN`)
  // instead of building 64-bit addresses with five scalar instructions each
  const char* list_c = reinterpret_cast<const char*>(list) + 4u * part;
  // the batch's first global index folded into the base (the loads only ever see base + offset of a real record)
  const char* base_c = reinterpret_cast<const char*>(S.rec32) - (size_t)first * (size_t)(4 * kRec32Stride[TYPE]);
  auto entry = [&](uint32_t k) { return *reinterpret_cast<const int*>(list_c + min(k, n - 1) * (4u * WPT)); };
  auto record = [&](int g) {
    return reinterpret_cast<const float*>(base_c + (uint32_t)g * (uint32_t)(4 * kRec32Stride[TYPE]));
  };
  auto field = [&](uint32_t k) { return min(ord0 + k * WPT + part + 1, ordmask); };
  int gA = entry(0);
  int gB = entry(1);
  A.load(record(gA));
  for (uint32_t i = 0; i < n; i += 2) {
    // B <- entry i+1 (clamped: reloading a valid record is harmless), then the list word of entry i+2
    B.load(record(gB));
    const int gNow = gA;
    gA = entry(i + 2);
    __builtin_amdgcn_sched_barrier(0);        // keep the loads above, the arithmetic below (hipcc would sink them)
    op(A, gNow, field(i));
    if (i + 1 >= n) break;
    __builtin_amdgcn_sched_barrier(0);
    A.load(record(gA));
    const int gNext = gB;
    gB = entry(i + 3);
    __builtin_amdgcn_sched_barrier(0);
    op(B, gNext, field(i + 1));
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same stream fed by VECTOR loads: lane l fetches list entry l of a 64-entry chunk and that entry's record
// (one coalesced list load, then kQuads 16-byte gathers per lane -- 64 records in flight at once, and the next chunk's
// while this one is evaluated); entry e's record then reaches the scalar registers by v_readlane.  Costs one readlane
// per record word instead of the scalar loads, but a record's latency is paid once per chunk, not once per entry.
#ifndef SRH_SWEEP_VEC
#define SRH_SWEEP_VEC 0
#endif
template <int TYPE, int WPT, class Op>
__device__ __forceinline__ void stream_list_vec(const SegDev& S, const uint32_t* __restrict__ list, uint32_t n_all,
                                                uint32_t ord0, uint32_t part, int lane, Op&& op, uint32_t ordmask = kOrdMask) {
  if (n_all <= part) return;
  constexpr int kQuads = kRec32Stride[TYPE] / 4;
  const uint32_t n = (n_all - part + WPT - 1) / WPT;
  const float4* base = reinterpret_cast<const float4*>(S.rec32);
  const int first = S.first;
  float4 cur[kQuads], nxt[kQuads];
  auto fetch = [&](uint32_t c0, float4 (&dst)[kQuads]) {
    const uint32_t k = min(c0 + (uint32_t)lane, n - 1);          // clamped: the spare lanes reload a valid record
    const int g = (int)list[k * WPT + part];
    const float4* r = base + (size_t)(g - first) * kQuads;
#pragma unroll
    for (int q = 0; q < kQuads; ++q) dst[q] = r[q];
  };
  fetch(0, cur);
  for (uint32_t c0 = 0; c0 < n; c0 += 64) {
    const bool more = c0 + 64 < n;
    if (more) fetch(c0 + 64, nxt);
    const uint32_t m = min(64u, n - c0);
#pragma unroll 1
    for (uint32_t e = 0; e < m; ++e) {
      RejectRecord<TYPE> A;
#pragma unroll
      for (int q = 0; q < kQuads; ++q) {
        A.v[4 * q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur[q].x), (int)e));
        A.v[4 * q + 1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur[q].y), (int)e));
        A.v[4 * q + 2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur[q].z), (int)e));
        A.v[4 * q + 3] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur[q].w), (int)e));
      }
      op(A, 0, min(ord0 + (c0 + e) * WPT + part + 1, ordmask));
    }
    if (more) {
#pragma unroll
      for (int q = 0; q < kQuads; ++q) cur[q] = nxt[q];
    }
  }
}

template <int TYPE, bool PRETEST, int WPT, bool DENKEYS>
__device__ __forceinline__ void sweep_list(const SegDev& S, const uint32_t* __restrict__ list, uint32_t n_all,
                                           uint32_t ord0, QuadState& Q, uint32_t part, int lane) {
#if SRH_SWEEP_VEC
  stream_list_vec<TYPE, WPT>(S, list, n_all, ord0, part, lane, [&](const RejectRecord<TYPE>& R, int, uint32_t field) {
    sweep_entry<TYPE, PRETEST, DENKEYS>(R, field, Q);
  }, Q.ordmask);
#else
  stream_list<TYPE, WPT>(S, list, n_all, ord0, part, [&](const RejectRecord<TYPE>& R, int, uint32_t field) {
    sweep_entry<TYPE, PRETEST, DENKEYS>(R, field, Q);
  }, Q.ordmask);
#endif
}

// Keys of the one-batch kernels of PLANAR primitives hold den = inv |D| instead of inv (see pair_bounds): one packed
// multiply per entry less, and it is the unit the matrix path produces.  Spheres carry a per-sphere inverse depth, and
// the all-types kernel mixes them with the others: those keep inv.
__host__ __device__ constexpr bool den_keys(int batch) {
  return batch == SRH_PRIM_DISK || batch == SRH_PRIM_TRIANGLE || batch == SRH_PRIM_PLANE;
}

// Re-sweep, lane-parallel: every lane whose pixel is still undecided (`open`) walks the tile's list again with the
// wave, and confirms in fp64 each candidate whose bound reaches the depth confirmed so far for ITS pixel.  The
// entry is wave-uniform, so the records arrive by scalar loads and the fp64 confirmation runs for all lanes that
// need it at once.  Used when several pixels of a round are undecided (overlapping coplanar splats, clouds of
// primitives without a usable depth estimate); a lone undecided pixel is cheaper on the wave-serial walk below.
template <int TYPE, bool PRETEST, bool TCH, int BATCH = -1>
__device__ __forceinline__ void resweep_list(const FrameDev& F, const SegDev& S, const uint32_t* __restrict__ list,
                                             uint32_t n, bool open, const f32x2 (&cf)[2], float rf,
                                             const f32x2 (&rlen)[2], const double d[3], float& bound, double& best,
                                             int& besti) {
  stream_list<TYPE, 1>(S, list, n, 0u, 0u, [&](const RejectRecord<TYPE>& R, int g, uint32_t) {
    int32_t sel[4];
    f32x2 inv[2];
    pair_bounds<TYPE, PRETEST>(R, cf, rf, rlen, sel, inv);
    const float iv = inv[0][0];
    const bool need = open && sel[0] && iv > 0.0f && iv >= reach_of(bound);
    if (__builtin_amdgcn_ballot_w64(need)) {
      if (need) {
        const double* R64 = S.rec64 + (size_t)(g - S.first) * kRec64Stride[TYPE];
        resolve_lex(F, hit_any64(TYPE, R64, F.o, d, TCH), g, best, besti);
        bound = float_above(best);
      }
    }
  });
}

// Slow path, one pixel at a time, the whole wave helping: the pixel's state is broadcast, lane l examines list
// entries l, l+64, ... (vector loads this time) and confirms those that are unbounded or whose bound reaches the
// pixel's confirmed depth; the per-lane results are merged with a lexicographic (t, index) minimum over the wave
// by cross-lane shuffles.
struct SlowPixel {
  f32x2 cf[2], rlen[2];   // the one pixel, replicated: the slow path reuses the sweep's quad arithmetic
  float rf;
  float bound;        // fp32 value >= the depth confirmed so far
  int g1, g2;         // already confirmed
};

template <int TYPE, bool PRETEST, bool TCH, int BATCH = -1>
__device__ __forceinline__ void slow_list(const FrameDev& F, const SegDev& S, const uint32_t* __restrict__ list,
                                          uint32_t n, int lane, const SlowPixel& P, const double d[3],
                                          double& best, int& besti) {
  for (uint32_t i = lane; i < n; i += 64) {
    const int g = (int)list[i];
    RejectRecord<TYPE> R;
    R.load(S.rec32 + (size_t)(g - S.first) * kRec32Stride[TYPE]);
    int32_t sel[4];
    f32x2 inv[2];
    pair_bounds<TYPE, PRETEST>(R, P.cf, P.rf, P.rlen, sel, inv);
    const float iv = inv[0][0];
    if (sel[0] && iv > 0.0f && iv >= reach_of(P.bound) && g != P.g1 && g != P.g2)
      confirm_global<TCH, BATCH>(F, g, d, best, besti);
  }
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __shfl_xor((unsigned)u, mask), hi = __shfl_xor((unsigned)(u >> 32), mask);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float readlane_f32(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

__device__ __forceinline__ double readlane_f64(double v, int src) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// The lists of a tile: per object batch its frame-wide `large` primitives (pass 0), then the tile's bin (pass 1).
struct TileLists {
  const FrameDev& F;
  int tile;
  __device__ __forceinline__ const uint32_t* list(int s, int pass) const {
    return pass == 0 ? F.large + F.seg[s].first : bin_list(F, s * F.ntiles_pad + tile);
  }
  __device__ __forceinline__ uint32_t count(int s, int pass) const {
#ifdef SRH_ABL_NOLOOP
    return 0;
#else
    const int bin = s * F.ntiles_pad + tile;
    return pass == 0 ? large_length(F, s) : bin_length(F, bin);
#endif
  }
};

// `passes`: bit 0 = the frame-wide lists, bit 1 = the tile's bins (a list that is left out still counts its ordinals)
template <bool PRETEST, int WPT, int BATCH = -1>
__device__ __forceinline__ void sweep_tile(const FrameDev& F, int tile, QuadState& Q, uint32_t part, int lane,
                                           int passes = 3) {
  const TileLists L{F, tile};
  uint32_t ord0 = 0;
  for (int s = 0; s < ((BATCH >= 0) ? 1 : F.nseg); ++s) {
    const SegDev& S = F.seg[s];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const uint32_t* list = L.list(s, pass);
      const uint32_t n = L.count(s, pass);
      if (!((passes >> pass) & 1)) { ord0 += n; continue; }
      constexpr bool DK = den_keys(BATCH);
      switch (BATCH >= 0 ? BATCH : S.type) {
        case SRH_PRIM_DISK: sweep_list<SRH_PRIM_DISK, PRETEST, WPT, DK>(S, list, n, ord0, Q, part, lane); break;
        case SRH_PRIM_PLANE: sweep_list<SRH_PRIM_PLANE, PRETEST, WPT, DK>(S, list, n, ord0, Q, part, lane); break;
        case SRH_PRIM_SPHERE: sweep_list<SRH_PRIM_SPHERE, PRETEST, WPT, DK>(S, list, n, ord0, Q, part, lane); break;
        default: sweep_list<SRH_PRIM_TRIANGLE, PRETEST, WPT, DK>(S, list, n, ord0, Q, part, lane); break;
      }
      ord0 += n;
    }
  }
}

// Global primitive index of the entry with ordinal `ord` (per lane) in the tile's lists; -1 if out of range.
template <int BATCH = -1>
__device__ __forceinline__ int ordinal_to_global(const FrameDev& F, int tile, uint32_t ord) {
  const TileLists L{F, tile};
  if (((BATCH >= 0) || F.nseg == 1) && L.count(0, 0) == 0) {          // one batch, nothing frame-wide: the ordinal is the bin-list position
    const uint32_t n = L.count(0, 1);
    return ord < n ? (int)L.list(0, 1)[min(ord, n - 1)] : -1;
  }
  int g = -1;
  bool done = false;
  for (int s = 0; s < ((BATCH >= 0) ? 1 : F.nseg); ++s) {
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const uint32_t n = L.count(s, pass);
      if (!done && ord < n) { g = (int)L.list(s, pass)[ord]; done = true; }
      if (!done) ord -= n;
    }
  }
  return g;
}

// Resolve the pixel of lane `src` on the slow path with the whole wave; returns the merged (t, index) minimum of
// everything confirmed here (index 0x7fffffff = nothing), valid in every lane.
template <bool PRETEST, bool TCH, int BATCH = -1>
__device__ __forceinline__ void slow_pixel(const FrameDev& F, int tile, int lane, int src, float cf, float rf,
                                           float len, float bound, int g1, int g2, const double d[3],
                                           double& out_t, int& out_i) {
  SlowPixel P;
  const float cs = readlane_f32(cf, src), ls = __builtin_amdgcn_rcpf(readlane_f32(len, src));
  P.cf[0] = P.cf[1] = f32x2{cs, cs};
  P.rlen[0] = P.rlen[1] = f32x2{ls, ls};
  P.rf = readlane_f32(rf, src);
  P.bound = readlane_f32(bound, src);
  P.g1 = __builtin_amdgcn_readlane(g1, src);
  P.g2 = __builtin_amdgcn_readlane(g2, src);
  const double ds[3] = {readlane_f64(d[0], src), readlane_f64(d[1], src), readlane_f64(d[2], src)};
  double best = __builtin_inf();
  int besti = 0x7fffffff;
  const TileLists L{F, tile};
  for (int s = 0; s < ((BATCH >= 0) ? 1 : F.nseg); ++s) {
    const SegDev& S = F.seg[s];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const uint32_t* list = L.list(s, pass);
      const uint32_t n = L.count(s, pass);
      switch (BATCH >= 0 ? BATCH : S.type) {
        case SRH_PRIM_DISK: slow_list<SRH_PRIM_DISK, PRETEST, TCH, BATCH>(F, S, list, n, lane, P, ds, best, besti); break;
        case SRH_PRIM_PLANE: slow_list<SRH_PRIM_PLANE, PRETEST, TCH, BATCH>(F, S, list, n, lane, P, ds, best, besti); break;
        case SRH_PRIM_SPHERE: slow_list<SRH_PRIM_SPHERE, PRETEST, TCH, BATCH>(F, S, list, n, lane, P, ds, best, besti); break;
        default: slow_list<SRH_PRIM_TRIANGLE, PRETEST, TCH, BATCH>(F, S, list, n, lane, P, ds, best, besti); break;
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {         // wavefront-shuffle lexicographic minimum of (t, index)
    const double ot = shfl_xor_f64(best, m);
    const int oi = __shfl_xor(besti, m);
    if (ot < best || (ot == best && oi < besti)) { best = ot; besti = oi; }
  }
  out_t = best;
  out_i = besti;
}

template <bool PRETEST, bool TCH, int BATCH = -1>
__device__ __forceinline__ void resweep_tile(const FrameDev& F, int tile, bool open, float cf, float rf, float len,
                                             const double d[3], float& bound, double& best, int& besti) {
  const f32x2 cfq[2] = {f32x2{cf, cf}, f32x2{cf, cf}};
  const float rl = __builtin_amdgcn_rcpf(len);
  const f32x2 rlq[2] = {f32x2{rl, rl}, f32x2{rl, rl}};
  const TileLists L{F, tile};
  for (int s = 0; s < ((BATCH >= 0) ? 1 : F.nseg); ++s) {
    const SegDev& S = F.seg[s];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const uint32_t* list = L.list(s, pass);
      const uint32_t n = L.count(s, pass);
      switch (BATCH >= 0 ? BATCH : S.type) {
        case SRH_PRIM_DISK: resweep_list<SRH_PRIM_DISK, PRETEST, TCH, BATCH>(F, S, list, n, open, cfq, rf, rlq, d, bound, best, besti); break;
        case SRH_PRIM_PLANE: resweep_list<SRH_PRIM_PLANE, PRETEST, TCH, BATCH>(F, S, list, n, open, cfq, rf, rlq, d, bound, best, besti); break;
        case SRH_PRIM_SPHERE: resweep_list<SRH_PRIM_SPHERE, PRETEST, TCH, BATCH>(F, S, list, n, open, cfq, rf, rlq, d, bound, best, besti); break;
        default: resweep_list<SRH_PRIM_TRIANGLE, PRETEST, TCH, BATCH>(F, S, list, n, open, cfq, rf, rlq, d, bound, best, besti); break;
      }
    }
  }
}

// per-pixel result of the sweep, parked in LDS between the sweep and the finish phase
struct alignas(16) Parked {
  int32_t k1, k2, k3, k4;
};

}  // namespace srh
#include "srh_mfma.h"
namespace srh {

// Workgroup -> tiles.  A workgroup renders 4 tiles that are neighbours in x (one per wave).  Workgroups are dealt to
// the 8 XCDs round-robin by the hardware, and each XCD has its own L2; a primitive overlaps neighbouring tiles, so
// the image is cut into REGIONS of kRegionW x kRegionH workgroups and whole regions are dealt to the XCDs in turn:
// a primitive's records are then fetched into one or two L2s instead of all eight.  The regions are small and the
// deal is rotated from one row of regions to the next (no XCD owns a vertical stripe), so every XCD gets an equal
// share of the busy parts of the image -- measured at 2048^2 x 100k discs: linear order 188 us, 128 x 64-pixel regions
// 152 us, unrotated 256-pixel stripes 190 us.
#ifndef SRH_REGION_W
#define SRH_REGION_W 2
#endif
#ifndef SRH_REGION_H
#define SRH_REGION_H 4
#endif
constexpr unsigned kRegionW = SRH_REGION_W, kRegionH = SRH_REGION_H;

__host__ __device__ inline unsigned binned_regions_x(const FrameDev& F) {
  return (((unsigned)F.tiles_x + 3u) / 4u + kRegionW - 1u) / kRegionW;
}

// number of 4-tile groups launched: whole regions, rounded up to a multiple of 8 regions
__host__ inline unsigned binned_grid(const FrameDev& F) {
  const unsigned tiles_y = (unsigned)F.tiles_y;
  const unsigned regions = binned_regions_x(F) * ((tiles_y + kRegionH - 1u) / kRegionH);
  return ((regions + 7u) & ~7u) * (kRegionW * kRegionH);
}

#ifndef SRH_SPLIT_TILES
#define SRH_SPLIT_TILES 3072
#endif
// Workgroup size of the one-wave-per-tile kernel.  Its waves never meet at a barrier, and tiles differ a lot in cost
// (10th / 50th / 90th percentile of a wave's life at config 5: 3 / 29 / 50 us): with four waves per workgroup a
// finished wave's slot stays empty until the workgroup's slowest wave is done and four slots are free together --
// measured 3.0 resident waves per SIMD of the 4 the registers allow.  One wave per workgroup: a slot is refilled
// as soon as its wave ends.  That only pays when the per-primitive kernels of the frames in flight use one-wave
// workgroups too (SRH_BIN_BLOCK above), or they starve behind the render waves.
#ifndef SRH_GROUP_WAVES
#define SRH_GROUP_WAVES 1
#endif
constexpr int kWavesPerGroup1 = SRH_GROUP_WAVES;   // 1 or 4
static_assert(kWavesPerGroup1 == 1 || kWavesPerGroup1 == 4, "SRH_GROUP_WAVES");
// waves per tile of the render kernel (see k_render_binned): 4 below SRH_SPLIT_TILES tiles, else 1
__host__ inline int binned_waves_per_tile(const FrameDev& F) { return F.ntiles < SRH_SPLIT_TILES ? 4 : 1; }

// lanes below this one in a ballot mask
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The finish phase is COMPACTED: after the sweep the tile's pixels that have at least one candidate are queued
// (row-major) in LDS and handed out 64 at a time, so a tile that is 40 % covered costs two fp64 rounds instead
// of four; pixels without a candidate are background and stored straight away.
// WPT = waves per tile.  1: a workgroup renders four tiles, one per wave (most work per launched wave).  4: a
// workgroup renders ONE tile -- each wave sweeps a quarter of the tile's entries for all 256 pixels, the per-pixel
// keys are merged through LDS, and each wave finishes a quarter of the pixels.  Same arithmetic, a quarter of the
// latency per tile and four times the waves: for frames (or row slabs of a multi-GPU job) with too few tiles to
// fill 1024 SIMDs several times over.
// F is restrict-qualified: in the many-views kernels it refers to device memory, and without the promise that no store
// of this function touches it every store would force the frame constants to be read again
// image / depth stores: written once and never read again by the GPU -> nontemporal (config 5: -1 % job time;
// sc1 write-through stores cost +6 %)
#ifndef SRH_OUT_NT
#define SRH_OUT_NT 1
#endif
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void out_store(float* p, float v) {
#if SRH_OUT_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void out_store4(float* p, float4 v) {
#if SRH_OUT_NT
  const vf4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<vf4*>(p));
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}

// Tile of a wave of the binned render grid (see "Workgroup -> tiles" above).  kWaves = waves of the workgroup.
template <int WPT, int kWaves>
__device__ __forceinline__ void binned_tile_of(const FrameDev& F, int wave, int& tx, int& ty) {
    // `group` = four tiles that are neighbours in x: one workgroup (WPT 1) or four consecutive ones on the same XCD
    const unsigned seq = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const unsigned idx = kWaves == 4 && WPT == 1 ? seq : seq >> 2;
    const unsigned sub = kWaves == 4 && WPT == 1 ? (unsigned)wave : (seq & 3u);
    const unsigned q = idx / (kRegionW * kRegionH), within = idx % (kRegionW * kRegionH);
    const unsigned nrx = binned_regions_x(F);
    const unsigned region = q * 8u + ((xcd + 3u * ((q * 8u) / nrx)) & 7u);   // rotate the deal from one region row to the next
    const unsigned rx = region % nrx, ry = region / nrx;
    tx = (int)((rx * kRegionW + within % kRegionW) * 4u + sub);
    ty = (int)(ry * kRegionH + within / kRegionW);
}

template <bool TCH, int WPT, int BATCH = -1>
__device__ __forceinline__ void render_binned_body(const FrameDev& __restrict__ F, float* __restrict__ image,
                                                   float* __restrict__ depth, int32_t* __restrict__ nearest) {
  constexpr int kWaves = WPT == 1 ? kWavesPerGroup1 : 4;   // waves of the workgroup
  __shared__ Parked park[kWaves][4][64];      // [wave][pixel of the quad][lane]: conflict-free 16-byte writes
  __shared__ int32_t front[kWaves][4][64];    // global index of each pixel's front candidate (-1: none / saturated)
  __shared__ uint8_t queue[kWaves][256];      // [wave]: ids j * 64 + lane of the pixels with a candidate, row-major
  const int wave = kWaves == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  int tx, ty;
  binned_tile_of<WPT, kWaves>(F, wave, tx, ty);
  // WPT 1: waves are independent, no block barrier below.  WPT 4: the whole workgroup shares the tile and leaves together.
  if (tx >= F.tiles_x || ty >= F.tiles_y) return;
  const int tile = ty * F.tiles_x + tx;
  const int px0 = tx * kTile, py0 = F.row0 + ty * kTile;
  const int c0 = px0 + 4 * (lane & 3);
  const int r_raw = py0 + (lane >> 2);
  const bool row_live = r_raw < F.row1;
#ifdef SRH_ABL_NOPRETEST
  const bool pretest = false;
#else
  const bool pretest = F.near_clip > 0.0;     // with near <= 0 a negative t can be valid: confirm every candidate
#endif
  const bool want_aux = F.normal_out || F.pos_out;
  const uint32_t mine = WPT == 1 ? 0xFu : (1u << wave);   // pixels of the quad this wave finishes
  uint32_t has = 0;                           // bit j: pixel j of the quad exists, is mine and has a candidate
  // a tile nobody reaches (every list empty; the same for all waves of a workgroup that shares the tile) goes straight
  // to the background stores: no ray set-up, no sweep
  uint32_t listed = 0;                        // entries of the tile's lists, all batches
  {
    const TileLists L{F, tile};
    for (int s = 0; s < ((BATCH >= 0) ? 1 : F.nseg); ++s) listed += L.count(s, 0) + L.count(s, 1);
  }
  // low key bits that hold a candidate's list position (ord_mask_for); the matrix-core measurement build packs its
  // keys with the full 12 bits, so it keeps them everywhere
  const uint32_t ordmask = SRH_MFMA ? kOrdMask : ord_mask_for(listed);
  if (listed) {
    const int r = min(r_raw, F.row1 - 1);
    QuadState Q;
    Q.ordmask = ordmask;
    Q.rf = (float)r;
    // |D|^2 of the quad's four pixels: D(c0 + j) = D(c0) + j Dc, so |D|^2 = A + j (B + j C) with A = |D(c0)|^2,
    // B = 2 D(c0).Dc, C = |Dc|^2 -- one fp64 ray instead of four (equal to ~1e-16 relative; only the fp32 bound uses it)
    double len2[4] = {1.0, 1.0, 1.0, 1.0};
    if (!den_keys(BATCH)) {
#ifdef SRH_ABL_LEN2_EACH
#pragma unroll
    for (int j = 0; j < 4; ++j) len2[j] = pixel_len2(F, min(c0 + j, F.W - 1), r);
#else
    {
      const double xs = c0 * F.step_x + -1.0, ys = (F.H > 1 && r == F.H - 1) ? -1.0 : (r * F.step_y + 1.0);
      const double X = xs * F.half_w, Y = ys * F.half_h, sx = F.step_x * F.half_w;
      double A = 0.0, B = 0.0, C = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double v = __builtin_fma(F.bx[i], X, __builtin_fma(F.by[i], Y, -F.bz[i] * F.focal)), dc = F.bx[i] * sx;
        A = __builtin_fma(v, v, A);
        B = __builtin_fma(v, dc, B);
        C = __builtin_fma(dc, dc, C);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) len2[j] = __builtin_fma((double)j, __builtin_fma((double)j, C, 2.0 * B), A);
    }
#endif
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = min(c0 + j, F.W - 1);
      Q.cf[j >> 1][j & 1] = (float)c;
      // 1 / |D| = rsq(|D|^2): the fp32 conversion (half an ulp of |D|^2) and v_rsq_f32 (1 ulp) together stay below
      // 1.3 * 2^-23 relative, well inside the 2^-20 the depth estimate reserves (plane_estimate_record)
      Q.rlen[j >> 1][j & 1] = __builtin_amdgcn_rsqf((float)len2[j]);
      Q.k1[j] = Q.k2[j] = Q.k3[j] = Q.k4[j] = kNoKey;
    }
    const uint32_t part = WPT == 1 ? 0u : (uint32_t)wave;     // which share of the tile's entries this wave sweeps
    // Disc bins go through the matrix cores (srh_mfma.h) unless the tile's lists are too long for an unsaturated key
    // field; the frame-wide list (huge discs, overflowing bins: usually empty) stays on the vector path, whose
    // centre-relative evaluation does not care how far the tile is from the ellipse.
    uint32_t n_wide = 0, n_bin = 0;
    bool matrix = false;
    if (SRH_MFMA && BATCH == SRH_PRIM_DISK) {
      const TileLists L{F, tile};
      n_wide = L.count(0, 0);
      n_bin = L.count(0, 1);
      matrix = n_wide + n_bin + 1u <= kOrdMask;
    }
    const int passes = matrix ? (n_wide ? 1 : 0) : 3;
    if (passes) {
      if (pretest) sweep_tile<true, WPT, BATCH>(F, tile, Q, part, lane, passes);
      else sweep_tile<false, WPT, BATCH>(F, tile, Q, part, lane, passes);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Parked p;
      p.k1 = Q.k1[j]; p.k2 = Q.k2[j]; p.k3 = Q.k3[j]; p.k4 = Q.k4[j];
      park[wave][j][lane] = p;
    }
    if (SRH_MFMA && BATCH == SRH_PRIM_DISK && matrix && n_bin > part) {
      int32_t K[8][4];
#pragma unroll
      for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) K[g][q] = kNoKey;
      const TileLists L{F, tile};
      if (pretest) sweep_bin_mfma<true, WPT>(F.seg[0], L.list(0, 1), n_bin, n_wide, part, lane, px0, py0, K);
      else sweep_bin_mfma<false, WPT>(F.seg[0], L.list(0, 1), n_bin, n_wide, part, lane, px0, py0, K);
      mfma_merge_keys<WPT>(lane, K);
      // both lanes of a pixel now hold its keys: lanes 0-31 file the even groups, lanes 32-63 the odd ones, into the
      // (row quad, lane) slots of the vector layout, on top of whatever the frame-wide list left there
      wave_lds_fence();
      const bool upper = lane >= 32;
      const int col = lane & 31, x = col & 15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = 2 * (2 * i + (upper ? 1 : 0)) + (col >> 4);
        Parked& slot = park[wave][x & 3][y * 4 + (x >> 2)];
        Parked p = slot;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int32_t key = upper ? K[2 * i + 1][q] : K[2 * i][q];
          if (kKeys == 4) p.k4 = imed3(p.k3, key, p.k4);
          p.k3 = imed3(p.k2, key, p.k3);
          p.k2 = imed3(p.k1, key, p.k2);
          p.k1 = max(p.k1, key);
        }
        slot = p;
      }
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Parked p = park[wave][j][lane];
        Q.k1[j] = p.k1; Q.k2[j] = p.k2; Q.k3[j] = p.k3; Q.k4[j] = p.k4;
      }
    }
    if (WPT > 1) {
      // merge: pixel j = wave of every lane collects the keys the four waves found for it
      __syncthreads();
      const int j = wave;
      int32_t m1 = kNoKey, m2 = kNoKey, m3 = kNoKey, m4 = kNoKey;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const Parked o = park[v][j][lane];
        const int32_t keys[4] = {o.k1, o.k2, o.k3, o.k4};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (kKeys == 4) m4 = imed3(m3, keys[t], m4);
          m3 = imed3(m2, keys[t], m3);
          m2 = imed3(m1, keys[t], m2);
          m1 = max(m1, keys[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t == j) { Q.k1[t] = m1; Q.k2[t] = m2; Q.k3[t] = m3; Q.k4[t] = m4; }
      Parked p;
      p.k1 = m1; p.k2 = m2; p.k3 = m3; p.k4 = m4;
      park[wave][j][lane] = p;                  // only this wave reads or writes column j = wave from here on
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (((mine >> j) & 1u) && Q.k1[j] != kNoKey && row_live && c0 + j < F.W) has |= 1u << j;
    // the front keys' list entries, looked up here so that the four loads are in flight together and the finish
    // rounds start from a global index instead of a dependent list read
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      front[wave][j][lane] =
          (((has >> j) & 1u) && !key_saturated(Q.k1[j], ordmask)) ? ordinal_to_global<BATCH>(F, tile, key_ordinal(Q.k1[j], ordmask)) : -1;
    }
  }

  // queue of the pixels with a candidate, row-major over the tile
  // (slot of a pixel = pixels with a candidate in the lanes below + those of this lane's quad before it: four ballots
  // and their v_mbcnt chains -- 16 vector instructions where a shuffle prefix sum took ~35 and six LDS round trips)
  int n1;
  {
    int pos = 0;
    n1 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned long long bj = __builtin_amdgcn_ballot_w64(((has >> j) & 1u) != 0u);
      pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bj >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bj, (uint32_t)pos));
      n1 += __popcll(bj);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((has >> j) & 1u) queue[wave][pos++] = (uint8_t)(j * 64 + lane);
  }

  // background: what the fragment stage gives an all-miss pixel (tonemap(0), +inf or far + 1, index 0)
  if (row_live && has != mine) {
    const size_t row = (size_t)(r_raw - F.row0);
    const float bg = tonemap_zero(F);
    const float bgz = background_depth(F, __builtin_inf());
    float* px = image + row * F.img_stride + 3 * (size_t)c0;
    float* dz = depth + row * F.depth_stride + c0;
    int32_t* nr = nearest ? nearest + row * F.near_stride + c0 : nullptr;
    const bool aligned = (((uintptr_t)px | (uintptr_t)dz | (uintptr_t)nr) & 15u) == 0;
    if (WPT == 1 && has == 0 && c0 + 3 < F.W && aligned) {
      const float4 v = make_float4(bg, bg, bg, bg);
      out_store4(px, v); out_store4(px + 4, v); out_store4(px + 8, v);
      out_store4(dz, make_float4(bgz, bgz, bgz, bgz));
      if (nr) *reinterpret_cast<int4*>(nr) = make_int4(0, 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((((mine & ~has) >> j) & 1u) && c0 + j < F.W) {
          out_store(px + 3 * j, bg); out_store(px + 3 * j + 1, bg); out_store(px + 3 * j + 2, bg);
          out_store(dz + j, bgz);
          if (nr) nr[j] = 0;
        }
    }
    if (want_aux) {
      const float zero[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((((mine & ~has) >> j) & 1u) && c0 + j < F.W) store_aux(F, row, c0 + j, zero);
    }
  }

  wave_lds_fence();                           // park / queue writes of other lanes
#pragma unroll 1
  for (int base = 0; base < n1; base += 64) {
    {
      const bool live = base + lane < n1;
      const int id = live ? (int)queue[wave][base + lane] : 0;
      const int j = id >> 6, src = id & 63;
      const int c = px0 + 4 * (src & 3) + j, r = py0 + (src >> 2);
      const Parked p = park[wave][j][src];
      const int gfront = front[wave][j][src];
      // The front candidate's record and material index are requested before the ray is set up, so that their
      // latency overlaps that arithmetic; nearly always this candidate is also the winner that gets shaded.
      FrontRecord<TCH, BATCH> fr;
      fr.fetch(F, gfront);
      double d[3];
      const float len = (float)pixel_ray(F, c, r, d);
      double best = __builtin_inf();
      int besti = 0x7fffffff;
      float bound = __builtin_inff();
      int g1 = -1, g2 = -1;
      bool slow = false;
#ifndef SRH_ABL_NOCONFIRM
      if (live) {
        // Confirm the front candidates in key order while their bound still reaches the confirmed depth (the
        // first one always; the next ones after a near miss at an ellipse edge or for nearly coplanar primitives).
        // A saturated ordinal does not identify its primitive: such a pixel confirms everything on the slow path.
        const int32_t keys[3] = {p.k1, p.k2, p.k3};
        const int32_t sentinel = kKeys == 4 ? p.k4 : p.k3;
        bool saturated = false;
        // a key still "reaches" the confirmed depth iff its inverse-depth bound is >= reach_of(bound)
        // (one reciprocal per confirmation instead of one per key; 0 while nothing is confirmed).  Keys in den units
        // (den_keys): den = inv |D|, so the threshold is scaled by |D| -- rounded DOWN (len is the fp32 of the fp64
        // length; 1 - 2^-21 covers that and the product), clamped again so that a key without an estimate still
        // reaches everything.
        const float key_unit = den_keys(BATCH) ? len * 0.9999995f : 1.0f;
        float reach = 0.0f;
#pragma unroll
        for (int q = 0; q < kKeys - 1; ++q) {
          const int32_t key = keys[q];
          if (key != kNoKey && !saturated && key_inv(key, ordmask) >= reach) {
            if (key_saturated(key, ordmask)) {
              saturated = true;
            } else {
              const int g = (q == 0) ? gfront : ordinal_to_global<BATCH>(F, tile, key_ordinal(key, ordmask));
              if (q == 0) g1 = g;
              if (q == 1) g2 = g;
              if (q == 0) resolve_lex(F, fr.hit(F, d), g, best, besti);
              else confirm_global<TCH, BATCH>(F, g, d, best, besti);
              bound = float_above(best);
              reach = fminf(reach_of(bound) * key_unit, kReachMax);   // bound = inf (a miss) gives 0 again
            }
          }
        }
        // the fourth key is only a bound: if it still reaches the confirmed depth, somebody unknown might too
        slow = saturated || (sentinel != kNoKey && key_inv(sentinel, ordmask) >= reach);
        if (slow) { g1 = g2 = -1; }           // the slow path re-confirms; cheaper than excluding three indices
      }
#endif
#ifndef SRH_ABL_NORESWEEP
      unsigned long long todo = __builtin_amdgcn_ballot_w64(slow);
      if (__popcll(todo) > kSerialSlowMax) {      // several undecided pixels: all of them at once, lane-parallel
        if (slow) { best = __builtin_inf(); besti = 0x7fffffff; bound = __builtin_inff(); }   // confirm everything that passes
        if (pretest) resweep_tile<true, TCH, BATCH>(F, tile, slow, (float)c, (float)r, len, d, bound, best, besti);
        else resweep_tile<false, TCH, BATCH>(F, tile, slow, (float)c, (float)r, len, d, bound, best, besti);
        todo = 0;
      }
      while (todo) {                          // wave-uniform loop over the lanes whose pixel needs the slow path
        const int sl = __builtin_ctzll(todo);
        todo &= todo - 1;
        double st;
        int si;
        if (pretest) slow_pixel<true, TCH, BATCH>(F, tile, lane, sl, (float)c, (float)r, len, bound, g1, g2, d, st, si);
        else slow_pixel<false, TCH, BATCH>(F, tile, lane, sl, (float)c, (float)r, len, bound, g1, g2, d, st, si);
        if (lane == sl && si != 0x7fffffff && (st < best || (st == best && (si < besti || besti == 0x7fffffff)))) {
          best = st;
          besti = si;
        }
      }
#endif
      if (live) {
        if (besti == 0x7fffffff) besti = 0;   // nothing hit: np.argmin of an all-inf column
        float rgb[3], aux[6];
#ifdef SRH_ABL_NOSHADE
        rgb[0] = rgb[1] = rgb[2] = (float)d[0] + __int_as_float(p.k1);
#else
        const ShadeHint hint = fr.hint();
        shade_pixel_t<TCH, BATCH>(F, d, best, besti, rgb, want_aux ? aux : nullptr, &hint);
#endif
        const size_t row = (size_t)(r - F.row0);
        float* px = image + row * F.img_stride + 3 * (size_t)c;
        out_store(px, rgb[0]); out_store(px + 1, rgb[1]); out_store(px + 2, rgb[2]);
        out_store(depth + row * F.depth_stride + c, background_depth(F, best));
        if (want_aux) store_aux(F, row, c, aux);
#ifdef SRH_DIAG_AGAIN   // diagnostic build: did this pixel take the slow path
        if (nearest) nearest[row * F.near_stride + c] = slow ? 1 : 0;
#else
        if (nearest) nearest[row * F.near_stride + c] = besti;
#endif
      }
    }
  }
  // Leave the counters as the next frame's binning needs them: zero -- a frame is then prep-and-bin + render, without a
  // clearing launch in front (SrhParams.counters_clean).  A tile's own bin counters were read by nobody else.  The
  // frame-wide list lengths are read by every tile until the kernel ends: tile 0 zeroes the OTHER set and makes it the
  // next frame's (kLargeNext; a ticket that finds the last tile to finish was measured first: 16 384 returning atomics
  // on one word made the frame 0.115 ms instead of 0.078).
  if (!F.keep_bins) {
    if (WPT > 1) __syncthreads();             // the four waves of the tile have all finished reading
    if ((WPT == 1 || wave == 0) && lane == 0) {
      for (int s = 0; s < ((BATCH >= 0) ? 1 : F.nseg); ++s) F.counters[kCounterPad + s * F.ntiles_pad + tile] = 0u;
      if (tile == 0) {
        const uint32_t other = (F.counters[kLargeNow] & 1u) ^ 1u;
        for (int s = 0; s < SRH_MAX_SEGMENTS; ++s) F.counters[4u * other + s] = 0u;
        F.counters[kLargeNext] = other;
      }
    }
  }
}

// Waves per SIMD the register allocator aims at.  The all-types kernel needs 128 VGPRs (4 waves); the one-batch
// instantiations need 87-116: disc / sphere fit 5 waves with 0-4 registers spilled, plane 6 with none, the triangle
// kernel would spill 10 at 5 waves and is slower there (config 4: 0.0553 -> 0.0562 ms), so it stays at 4.  With 5
// waves per SIMD config 5 goes from 0.0927 to 0.0852 ms per frame (stand-alone kernel 90.5 -> 88.0 us): the extra
// slot is what the other frames' binning waves need beside the render waves.
#ifndef SRH_TYPED_WAVES
#define SRH_TYPED_WAVES 0
#endif
constexpr int typed_waves(bool tch, int batch) {      // the Phong fragment stage needs more registers: 4 waves as before
  return (batch < 0 || tch) ? 4 : SRH_TYPED_WAVES ? SRH_TYPED_WAVES
       : batch == SRH_PRIM_TRIANGLE ? 4 : batch == SRH_PRIM_PLANE ? 6 : 5;
}
template <bool TCH, int WPT, int BATCH = -1>
__global__ __launch_bounds__(WPT == 1 ? 64 * kWavesPerGroup1 : 256)
__attribute__((amdgpu_waves_per_eu(typed_waves(TCH, BATCH)))) void k_render_binned(
    FrameDev F, float* __restrict__ image, float* __restrict__ depth, int32_t* __restrict__ nearest) {
  render_binned_body<TCH, WPT, BATCH>(F, image, depth, nearest);
}

// The product form of the same kernel reads its frame constants from MEMORY in the constant address space -- the copy
// k_prep (or k_put_frame) left in the workspace, FrameDev::self -- instead of from by-value kernel arguments: the loads
// are invariant scalar loads that hipcc re-materialises where the values are used, where the by-value form keeps ~150
// scalars alive across the sweep and spills them into vector-register lanes, and every v_writelane / v_readlane of
// those is a VECTOR instruction (disc kernel: 219 of them in the by-value form, 56 here; 52 -> 22 spilled scalars,
// 96 -> 93 vector registers).  -DSRH_FRAME_MEM=0 builds the by-value launch for comparison.
#ifndef SRH_FRAME_MEM
#define SRH_FRAME_MEM 1
#endif
typedef const __attribute__((address_space(4))) FrameDev* FrameConstPtr;
template <bool TCH, int WPT, int BATCH = -1>
__global__ __launch_bounds__(WPT == 1 ? 64 * kWavesPerGroup1 : 256)
__attribute__((amdgpu_waves_per_eu(typed_waves(TCH, BATCH)))) void k_render_binned_mem(
    FrameConstPtr Fp, float* __restrict__ image, float* __restrict__ depth, int32_t* __restrict__ nearest) {
  render_binned_body<TCH, WPT, BATCH>(*(const FrameDev*)Fp, image, depth, nearest);
}

// A frame's constants into its workspace slot: one wave; used when the call that renders did not run k_prep itself
// (SRH_STAGE_RENDER alone)
__global__ __launch_bounds__(64) void k_put_frame(FrameDev F) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&F);
  uint32_t* dst = reinterpret_cast<uint32_t*>(F.self);
  for (unsigned i = threadIdx.x; i < sizeof(FrameDev) / 4; i += 64) dst[i] = src[i];
}

// ---- many views per launch (srh_render_views): blockIdx.y selects the view, whose FrameDev lives in device memory;
// the views' outputs are stacked (V, rows, W, .) -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_views_zero(const FrameDev* __restrict__ Fs) {
  const FrameDev& F = Fs[blockIdx.y];
  const size_t n = (size_t)kCounterPad + (size_t)F.nbins;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) F.counters[i] = 0u;
}
__global__ __launch_bounds__(kBinBlock) void k_bin_count_views(const FrameDev* __restrict__ Fs) { bin_count_body(Fs[blockIdx.y]); }

// The render kernel reads its view's frame constants from CONSTANT memory: loads from there are invariant, so hipcc
// re-materialises them where they are used (as it does with a by-value kernel argument) instead of keeping 140 SGPRs
// alive -- and spilling them -- across the whole kernel, which is what a plain device pointer costs (141 vs 114 us per
// frame at config 5).
constexpr int kMaxViewsPerCall = 256;
constexpr int kViewRing = 4;                       // batches in flight before the host has to wait
__constant__ FrameDev g_view_frames[kViewRing * kMaxViewsPerCall];

template <bool TCH, int WPT, int BATCH = -1>
__global__ __launch_bounds__(WPT == 1 ? 64 * kWavesPerGroup1 : 256)
__attribute__((amdgpu_waves_per_eu(typed_waves(TCH, BATCH)))) void k_render_binned_views(
    int base, float* __restrict__ image, float* __restrict__ depth, int32_t* __restrict__ nearest) {
  const FrameDev& F = g_view_frames[base + blockIdx.y];
  const size_t rows = (size_t)(F.row1 - F.row0), v = blockIdx.y;
  render_binned_body<TCH, WPT, BATCH>(F, image + v * rows * F.img_stride, depth + v * rows * F.depth_stride,
                                      nearest ? nearest + v * rows * F.near_stride : nullptr);
}

}  // namespace srh

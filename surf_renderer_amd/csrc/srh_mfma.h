// Matrix-core sweep of a disc bin (gfx950): the screen-space reject and the depth estimate of 32 list entries x 32 tile
// pixels per pair of MFMA instructions, beside the vector pipe.
//
// The vector sweep (srh_binned.h: pair_bounds) spends 16 of its 43 instructions per entry on two polynomials in the
// pixel coordinates: the ellipse form q (candidate iff q < 0) and the affine estimate den.  In TILE-LOCAL coordinates
// x, y in [0, 15] they are products of small matrices,
//     q(entry, pixel)   = [a0 a1 a2 a3 a4 a5] . [1 x y x^2 xy y^2]        den(entry, pixel) = [b0 b1 b2] . [1 x y]
// and the monomials are integers <= 225: exact in float16 and in bfloat16.  The fp32 matrix instructions do not help
// (v_mfma_f32_32x32x2_f32 runs at the vector rate and does not overlap with vector work: tools/ubench_mfma_coissue.hip,
// profiles/r03_ubench_mfma_coissue.txt) -- but v_mfma_f32_32x32x16_{f16,bf16} has K = 16, enough for the monomials
// times a SPLIT of every coefficient into 2 float16 parts (q: only its sign is used, so each entry is scaled by a power
// of two into float16's range; 22 bits) or 3 bfloat16 parts (den: 24 bits, fp32's own range), products are exact and
// sums are fp32: one instruction per polynomial per 32 x 32 block, 16 of 2 x 32 cycles of the SIMD's vector issue.
// The error budget of this evaluation is part of the records' margins (srh_reject.h: conic_record, plane_estimate_record).
//
// Layout (verified by tools/probe_mfma_layout.hip, profiles/r03_probe_mfma_layout.txt):  D = A (32 x 16) B (16 x 32),
//   A: lane l, slot s -> A[row l % 32][k = 8 (l / 32) + s]          rows    = list entries of the chunk
//   B: lane l, slot s -> B[k = 8 (l / 32) + s][column l % 32]       columns = pixels of a group (two tile rows)
//   D: lane l, register v -> D[row 8 (v / 4) + 4 (l / 32) + v % 4][column l % 32]
// so a lane ends up with ONE pixel and 16 entries: the four keys of that pixel stay in the lane's registers, and the
// two lanes l, l + 32 that share a pixel merge their keys once per tile.
//   K slots of q:    lanes  0-31: [ 1, x, x^2, 1, x, x^2, -, -] . [a0h a1h a3h a0l a1l a3l]
//                    lanes 32-63: [ y, xy, y^2, y, xy, y^2, -, -] . [a2h a4h a5h a2l a4l a5l]
//   K slots of den:  lanes  0-31: [ 1, x, 1, x, 1, x, -, -] . [b0h b1h b0m b1m b0l b1l]
//                    lanes 32-63: [ y, y, y, -, -, -, -, -] . [b2h b2m b2l]
// Every lane's monomials are polynomials in the group index: they are stepped with additions, exactly, up through the
// eight groups for one chunk of entries and down again for the next.
#pragma once

namespace srh {

// MEASURED AND NOT ADOPTED (round 3; DESIGN.md "measured and dropped"): bit-identical on the whole GPU suite, and
// slower -- config 5 at 0.097 ms per frame (four waves per SIMD, 58 spilled registers) and 0.130 (five waves, 216
// spilled) against 0.075 for the vector sweep.  The two matrix instructions do replace 16 vector instructions per entry,
// but what is left per (entry, pixel) value -- mask, sign select, key, four-key insertion: 7 instructions -- is the same
// work as before on one value at a time instead of packed pairs, the operands cost ~50 instructions per group once the
// register allocator starts re-deriving them, chunks of 32 entries round every list up, and 32 accumulator + 32 key
// registers leave no room: 174 vector instructions per 32 x 32 block against 172 on the vector path.  Kept as a
// measurement build: -DSRH_MFMA=1.
#ifndef SRH_MFMA
#define SRH_MFMA 0
#endif

typedef _Float16 mf_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 mf_b8 __attribute__((ext_vector_type(8)));
typedef float mf_acc __attribute__((ext_vector_type(16)));
typedef uint32_t mf_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mf_pack_h(float lo, float hi) {      // two floats -> packed float16 (round to nearest)
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 p = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ float mf_round_h(float v) { return (float)(_Float16)v; }
__device__ __forceinline__ uint32_t mf_bf_bits(float v) {                // bfloat16 (round to nearest) of v, in the low 16 bits
  return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v);
}
__device__ __forceinline__ float mf_bf_value(uint32_t bits) { return __uint_as_float(bits << 16); }

// v = h + m + l up to 2^-24 |v|, each part a bfloat16
__device__ __forceinline__ void mf_split3(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = mf_bf_bits(v);
  const float r1 = v - mf_bf_value(h);
  m = mf_bf_bits(r1);
  l = mf_bf_bits(r1 - mf_bf_value(m));
}

// (bits(den) & ~0xFFF) | field.  Plain C, not inline assembly: `den` comes straight out of an MFMA, and the wait states
// between a matrix instruction and a vector read of its result are inserted by the compiler -- for instructions it can
// see (an inline-assembly v_bfi_b32 here read registers the MFMA had not written yet: frames differed from run to run).
__device__ __forceinline__ int32_t mf_pack_key(uint32_t maskv, float den, uint32_t field) {
  return (int32_t)((__float_as_uint(den) & maskv) | field);
}

// The eight pixel groups of one chunk of entries: per group two matrix instructions, then the 16 (entry, pixel) values
// of this lane go into the pixel's keys.  DOWN = walk the groups 7 .. 0 (the monomials arrive at group 7's values).
template <bool DOWN, int WPT>
__device__ __forceinline__ void mfma_groups(const mf_u4& Aq, const mf_u4& Ad, uint32_t fbase, uint32_t maskv, uint32_t M1,
                                            uint32_t M2, float& m0, float& m1, float& m2, float dm0, float dm1, float& dm2,
                                            float ddm2, float& w, int32_t (&K)[8][4]) {
  const mf_acc zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    constexpr int kLast = 7;
    const int g = DOWN ? kLast - i : i;
    mf_u4 Bq, Bd;
    Bq[0] = mf_pack_h(m0, m1);
    Bq[1] = mf_pack_h(m2, m0);
    Bq[2] = mf_pack_h(m1, m2);
    Bq[3] = 0u;
    Bd[0] = mf_bf_bits(m0) | (mf_bf_bits(w) << 16);
    Bd[1] = Bd[0] & M1;
    Bd[2] = Bd[0] & M2;
    Bd[3] = 0u;
    const mf_acc q = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf_h8, Aq), __builtin_bit_cast(mf_h8, Bq), zero, 0, 0, 0);
    const mf_acc d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mf_b8, Ad), __builtin_bit_cast(mf_b8, Bd), zero, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const uint32_t field = fbase + (uint32_t)(WPT * (8 * (v / 4) + (v % 4)));
      const int32_t key = mf_pack_key(maskv, d[v], field) & (__float_as_int(q[v]) >> 31);
      if (kKeys == 4) K[g][3] = imed3(K[g][2], key, K[g][3]);
      K[g][2] = imed3(K[g][1], key, K[g][2]);
      K[g][1] = imed3(K[g][0], key, K[g][1]);
      K[g][0] = max(K[g][0], key);
    }
    if (i < 7) {
      if (!DOWN) { m0 += dm0; m1 += dm1; m2 += dm2; dm2 += ddm2; w += dm0; }
      else { dm2 -= ddm2; m2 -= dm2; m1 -= dm1; m0 -= dm0; w -= dm0; }
    }
  }
}

// One disc bin through the matrix cores.  K[g][0..3] = the four largest keys (descending) of the pixel this lane holds in
// group g (tile rows 2g, 2g + 1; x = lane % 16, y = 2g + (lane % 32) / 16), over the entries of this lane's half; see
// mfma_merge_keys.  `list` / n_all / ord0 / part as in stream_list.  Requires ord0 + n_all + 1 <= kOrdMask (the caller
// sends longer lists through the vector sweep, whose field saturates).
template <bool PRETEST, int WPT>
__device__ __forceinline__ void sweep_bin_mfma(const SegDev& S, const uint32_t* __restrict__ list, uint32_t n_all,
                                               uint32_t ord0, uint32_t part, int lane, int px0, int py0,
                                               int32_t (&K)[8][4]) {
  if (n_all <= part) return;
  const uint32_t n = (n_all - part + WPT - 1) / WPT;
  const bool upper = lane >= 32;
  const int col = lane & 31;
  const float xf = (float)(col & 15), y0f = (float)(col >> 4);
  // this lane's monomials at group 0 and their steps to the next group (y -> y + 2); lanes 0-31 hold constants
  float m0 = upper ? y0f : 1.0f, m1 = upper ? xf * y0f : xf, m2 = upper ? y0f * y0f : xf * xf;
  const float dm0 = upper ? 2.0f : 0.0f, dm1 = upper ? 2.0f * xf : 0.0f, ddm2 = upper ? 8.0f : 0.0f;
  float dm2 = upper ? 4.0f * y0f + 4.0f : 0.0f;
  float w = upper ? y0f : xf;                                   // second slot of den's first pair: x, or y again
  const uint32_t M1 = upper ? 0x0000FFFFu : 0xFFFFFFFFu, M2 = upper ? 0u : 0xFFFFFFFFu;
  const uint32_t maskv = 0xFFFFF000u;
  const float X0 = (float)px0, Y0 = (float)py0;
  const float4* rec = reinterpret_cast<const float4*>(S.rec32) - (size_t)S.first * 3;
  const uint32_t nchunks = (n + 31u) / 32u;
  for (uint32_t chunk = 0; chunk < nchunks; ++chunk) {
    // ---- operand A: tile-local coefficients of entry chunk * 32 + col, split for this lane's K half
    const uint32_t k = chunk * 32u + (uint32_t)col;
    const bool valid = k < n;
    const int gi = (int)list[part + WPT * min(k, n - 1u)];
    const float4 r0 = rec[(size_t)gi * 3], r1 = rec[(size_t)gi * 3 + 1];
    const float dx = X0 - r0.x, dy = Y0 - r0.y;
    const float a3 = r0.z, a4 = r0.w, a5 = r1.x;
    const float lin = __builtin_fmaf(a3, dx, a4 * dy);                         // A11 dx + 2A12 dy
    float a1 = __builtin_fmaf(a3, dx, lin);                                   // 2 A11 dx + 2A12 dy
    float a2 = __builtin_fmaf(a4, dx, 2.0f * (a5 * dy));                      // 2A12 dx + 2 A22 dy
    float a0 = __builtin_fmaf(dx, lin, __builtin_fmaf(a5 * dy, dy, -1.00000024f));
    float c0 = upper ? a2 : a0, c1 = upper ? a4 : a1, c2 = upper ? a5 : a3;
    // one power of two per entry brings the largest coefficient to [2^12, 2^13): only the sign of q is used
    const float big = fmaxf(fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fmaxf(fabsf(a2), fabsf(a3))), fmaxf(fabsf(a4), fabsf(a5)));
    const float sc = __builtin_amdgcn_ldexpf(1.0f, 13 - __builtin_amdgcn_frexp_expf(big));
    c0 *= sc; c1 *= sc; c2 *= sc;
    if (!valid) { c0 = upper ? 0.0f : 1.0f; c1 = 0.0f; c2 = 0.0f; }          // past the end of the list: q = 1, never a candidate
    const float h0 = mf_round_h(c0), h1 = mf_round_h(c1), h2 = mf_round_h(c2);
    mf_u4 Aq;
    Aq[0] = mf_pack_h(h0, h1);
    Aq[1] = mf_pack_h(h2, c0 - h0);
    Aq[2] = mf_pack_h(c1 - h1, c2 - h2);
    Aq[3] = 0u;
    mf_u4 Ad;
    {
      float b0 = __builtin_fmaf(Y0, r1.w, __builtin_fmaf(X0, r1.z, r1.y)), b1 = r1.z, b2 = r1.w;
      if (!PRETEST) { b0 = kNoEstimate; b1 = 0.0f; b2 = 0.0f; }              // near <= 0: every candidate ranks first
      uint32_t p0h, p0m, p0l, p1h, p1m, p1l;
      mf_split3(upper ? b2 : b0, p0h, p0m, p0l);
      mf_split3(b1, p1h, p1m, p1l);
      Ad[0] = p0h | ((upper ? p0m : p1h) << 16);
      Ad[1] = upper ? p0l : (p0m | (p1m << 16));
      Ad[2] = upper ? 0u : (p0l | (p1l << 16));
      Ad[3] = 0u;
    }
    const uint32_t fbase = __builtin_amdgcn_readfirstlane(ord0 + part + WPT * (32u * chunk) + 1u);
    // the monomials walk up through the groups for even chunks and back down for odd ones: two straight-line bodies,
    // chosen once per chunk (one body with the direction tested inside cost a branch per value)
    if (chunk & 1u) mfma_groups<true, WPT>(Aq, Ad, fbase, maskv, M1, M2, m0, m1, m2, dm0, dm1, dm2, ddm2, w, K);
    else mfma_groups<false, WPT>(Aq, Ad, fbase, maskv, M1, M2, m0, m1, m2, dm0, dm1, dm2, ddm2, w, K);
  }
}

// The two lanes l, l + 32 hold one pixel's keys over different halves of every chunk's rows: the upper half's rows lie
// 4 (x WPT list positions) further on, which its key fields have not counted yet.  After this both lanes hold the
// pixel's four largest keys.
template <int WPT>
__device__ __forceinline__ void mfma_merge_keys(int lane, int32_t (&K)[8][4]) {
  const bool upper = lane >= 32;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    int32_t mine[4], other[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mine[r] = (upper && K[g][r] != kNoKey) ? K[g][r] + 4 * WPT : K[g][r];
      other[r] = __shfl_xor(mine[r], 32);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int32_t key = other[r];
      if (kKeys == 4) mine[3] = imed3(mine[2], key, mine[3]);
      mine[2] = imed3(mine[1], key, mine[2]);
      mine[1] = imed3(mine[0], key, mine[1]);
      mine[0] = max(mine[0], key);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) K[g][r] = mine[r];
  }
}

}  // namespace srh

// Tiling shared by the direct first-layer kernels (conv0.hip fp32, conv0_b16.hip bf16-split):
// 256-output-pixel tiles, the raw uint8 input rows of a tile staged in LDS as (at most) two
// contiguous byte ranges, register prefetch of the next tile's patch.
#pragma once
#include "igemm.hpp"

namespace dx {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// native vectors: arrays of HIP's uint4 / float4 structs handed to a function by reference stay
// in scratch memory (every prefetched word round-trips through a scratch store + load)
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTile = 256;    // output pixels per tile

struct Seg {       // the (at most two) images a tile touches
  int cnt_a, cnt_b;      // pixels of the tile in the first / second image
  int pa;                // first pixel inside the first image
  int oya0;              // first output row of segment a
  long long src_a, src_b;  // byte offsets of the staged ranges in the observation buffer
  int bytes_a, bytes_b;
};

// The (at most two) images a tile touches, through the minibatch's gather table.  The table is
// read through the CONSTANT address space (never written by these kernels): a scalar load the
// compiler waits for with lgkmcnt where the value is first used.  (As a vector load it is waited
// for with vmcnt(0), which also drains every store of the previous tile; as a scalar load waited
// for on the spot it costs a full memory round trip per tile -- 1,800 of the forward's 12,000
// cycles per tile in round 2 -- so the kernels fetch a tile's pair one tile AHEAD.)
struct TileImages { int img_a, raw_a, raw_b; };

__device__ __forceinline__ TileImages tile_images(const Conv0Args &a, int m0) {
  typedef const __attribute__((address_space(4))) int32_t *ConstTable;
  const int P = a.h0 * a.w0;
  const int img_a = __builtin_amdgcn_readfirstlane(m0 / P);
  const bool gather = a.idx != nullptr;
  // no branch around the loads (a join would make the compiler wait for them on the spot): without
  // a table they read two words of the frame buffer that tile_segments ignores
  ConstTable tab = gather ? (ConstTable)(a.idx) + img_a : (ConstTable)(a.obs);
  const int last = (a.M - 1) / P;  // (the entry behind the last image is not touched)
  TileImages t;
  t.img_a = img_a;
  t.raw_a = tab[0];
  t.raw_b = tab[gather && img_a < last ? 1 : 0];
  return t;
}

__device__ __forceinline__ Seg tile_segments(const Conv0Args &a, int m0, const TileImages &im) {
  Seg s;
  const int P = a.h0 * a.w0;
  const int rowB = a.in_w * 4;
  const long long imgB = static_cast<long long>(a.in_h) * rowB;
  const int left = min(kTile, a.M - m0);
  const int img_a = m0 / P;
  s.pa = m0 - img_a * P;
  s.cnt_a = min(left, P - s.pa);
  s.cnt_b = left - s.cnt_a;
  s.oya0 = s.pa / a.w0;
  const int oya1 = (s.pa + s.cnt_a - 1) / a.w0;
  const long long ia = a.idx ? im.raw_a : im.img_a, ib = a.idx ? im.raw_b : im.img_a + 1;
  s.src_a = ia * imgB + static_cast<long long>(4 * s.oya0) * rowB;
  s.bytes_a = (4 * (oya1 - s.oya0) + 8) * rowB;
  s.src_b = 0;
  s.bytes_b = 0;
  if (s.cnt_b > 0) {
    s.src_b = ib * imgB;
    s.bytes_b = (4 * ((s.cnt_b - 1) / a.w0) + 8) * rowB;
  }
  return s;
}

__device__ __forceinline__ Seg tile_segments(const Conv0Args &a, int m0) {
  return tile_segments(a, m0, tile_images(a, m0));
}

// byte offset inside the LDS patch of input pixel (4*oy, 4*ox) for tile pixel p (0 if p is padding)
__device__ __forceinline__ int pixel_base(const Conv0Args &a, const Seg &s, int p) {
  const int rowB = a.in_w * 4;
  if (p < s.cnt_a) {
    const int pix = s.pa + p;
    const int oy = pix / a.w0, ox = pix - oy * a.w0;
    return 4 * (oy - s.oya0) * rowB + ox * 16;
  }
  if (p < s.cnt_a + s.cnt_b) {
    const int pix = p - s.cnt_a;
    const int oy = pix / a.w0, ox = pix - oy * a.w0;
    return s.bytes_a + 4 * oy * rowB + ox * 16;
  }
  return 0;
}

// Both byte ranges -> LDS patch (range b follows range a), split in two halves so that the
// global loads of tile t+1 are in flight during the MFMA body of tile t:
//   patch_load : up to 6 x 16 B per lane into registers, unconditional (clamped index)
//   patch_store: registers -> LDS after the barrier that retires the previous tile's reads
constexpr int kPatchRegs = 6;  // 6 * 256 lanes * 16 B = 24 KB >= the largest patch (launch-checked)

// NT = threads of the workgroup: 256 (6 registers per lane) or 512 (3)
template <int NT = 256>
__device__ __forceinline__ void patch_load(const Conv0Args &a, const Seg &s, u32x4 (&v)[kPatchRegs * 256 / NT]) {
  const u32x4 *sa = reinterpret_cast<const u32x4 *>(a.obs + s.src_a);
  const u32x4 *sb = reinterpret_cast<const u32x4 *>(a.obs + s.src_b);
  const int na = s.bytes_a / 16, n = na + s.bytes_b / 16;
#pragma unroll
  for (int u = 0; u < kPatchRegs * 256 / NT; ++u) {
    const int i = u * NT + threadIdx.x;
    const int ic = i < n ? i : 0;
    v[u] = *(ic < na ? sa + ic : sb + (ic - na));
  }
}

template <int NT = 256>
__device__ __forceinline__ void patch_store(const Seg &s, const u32x4 (&v)[kPatchRegs * 256 / NT], uint8_t *patch) {
  u32x4 *dst = reinterpret_cast<u32x4 *>(patch);
  const int n = (s.bytes_a + s.bytes_b) / 16;
#pragma unroll
  for (int u = 0; u < kPatchRegs * 256 / NT; ++u) {
    const int i = u * NT + threadIdx.x;
    if (i < n) dst[i] = v[u];
  }
}

inline int patch_bytes(const Conv0Args &a) {
  // rows of the (at most two) staged ranges: 4 per output row touched + 4 per range
  const int out_rows = kTile / a.w0 + 3;
  return (4 * out_rows + 8) * a.in_w * 4;
}

}  // namespace
}  // namespace dx

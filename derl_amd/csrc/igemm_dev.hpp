// Device-side helpers shared by the implicit-GEMM kernels (row decode, raw loads, dequant).
#pragma once
#include "igemm.hpp"

namespace dx {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// uint8 -> float / 255, bit-identical to IEEE division for every byte value: q = x*r, one
// fma residual step (checked exhaustively on the host in tests/test_host_logic.py).
__device__ __forceinline__ float dequant_u8(uint32_t x) {
  const float r = 1.0f / 255.0f;
  const float xf = static_cast<float>(x);
  const float q = xf * r;
  const float e = __builtin_fmaf(-q, 255.0f, xf);
  return __builtin_fmaf(e, r, q);
}

// Loads are unconditional (an invalid element reads offset 0 of the buffer and is zeroed by a
// select when the tile is written to LDS) and keep the RAW words in registers: conversion and
// masking happen one K-step later, after the MFMAs the load latency hides under.  A
// conditional load compiles to a branch with an immediate vmcnt(0), which serialises every
// load and leaves nothing in flight during the MFMAs.
template <bool U8> struct Raw { using type = float4; };
template <> struct Raw<true> { using type = uint32_t; };

template <bool U8>
__device__ __forceinline__ typename Raw<U8>::type load_raw(const void *base, long long off) {
  if constexpr (U8)
    return *reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(base) + off);
  else
    return *reinterpret_cast<const float4 *>(static_cast<const float *>(base) + off);
}

__device__ __forceinline__ float4 to_float4(float4 v) { return v; }
__device__ __forceinline__ float4 to_float4(uint32_t w) {
  return make_float4(dequant_u8(w & 0xff), dequant_u8((w >> 8) & 0xff), dequant_u8((w >> 16) & 0xff),
                     dequant_u8(w >> 24));
}
__device__ __forceinline__ float4 masked(float4 v, bool ok) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// One LDS-DMA piece: 64 lanes x 16 bytes to 1 KiB of LDS at `dst` (uniform), from `base` (UNIFORM)
// plus a per-lane 32-bit byte offset: the SGPR-base form of global_load_lds, which needs no vector
// ALU instruction for the address.  A 64-bit per-lane address costs a v_lshl_add_u64 per piece, and
// a VALU instruction issued beside waves that keep the matrix pipe full waits for a gap there.
// The empty asm keeps the compiler from widening the offset ahead of the loop (it then adds the
// base per piece with VALU after all).  Also: the builtin must not be spelled inside a kernel
// TEMPLATE with template-dependent arguments -- hipcc's host pass then silently emits no stub.
__device__ __forceinline__ void dma_piece(const void *base, uint32_t lane_bytes, float *dst) {
  // both operands opaque: the compiler otherwise folds several pieces' bases into one plus per-piece
  // constants and adds those to a widened lane offset with v_lshl_add_u64.  `base` MUST be uniform
  // (the "s" constraint would take lane 0's value otherwise).
  asm volatile("" : "+s"(base));
  asm volatile("" : "+v"(lane_bytes));
  __builtin_amdgcn_global_load_lds(static_cast<const char *>(base) + lane_bytes, dst, 16, 0, 0);
}

struct RowPos {
  long long base;   // element offset of the row's origin pixel (channel 0)
  uint32_t okmask;  // bit s: run s of this row lies inside the image (all runs if !check)
};

__device__ __forceinline__ RowPos decode_row(const Gather &g, int m, bool valid) {
  RowPos r;
  const uint32_t mm = valid ? static_cast<uint32_t>(m) : 0u;
  uint32_t img = fdiv(mm, g.div_img);
  const uint32_t rem = mm - img * g.OHW;
  const uint32_t oy = fdiv(rem, g.div_row);
  const uint32_t ox = rem - oy * g.OW;
  if (g.idx) img = static_cast<uint32_t>(g.idx[img]);
  const int y0 = static_cast<int>(oy) * g.sy, x0 = static_cast<int>(ox) * g.sx;
  r.base = static_cast<long long>(img) * g.img_stride + static_cast<long long>(y0 * g.W + x0) * g.C;
  uint32_t mask = 0xffffffffu;
  if (g.check) {  // uniform branch, prologue / once per row
    mask = 0;
    for (int s = 0; s < g.nseg; ++s) {
      const int yy = y0 + g.seg_dy[s], xx = x0 + g.seg_dx[s];
      const bool in = static_cast<unsigned>(yy) < static_cast<unsigned>(g.H) &&
                      static_cast<unsigned>(xx) < static_cast<unsigned>(g.W);
      mask |= (in ? 1u : 0u) << s;
    }
  }
  r.okmask = valid ? mask : 0u;
  return r;
}


}  // namespace
}  // namespace dx

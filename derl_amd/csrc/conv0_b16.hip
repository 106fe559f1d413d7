// First convolution (8x8 stride 4, uint8 NHWC frames -> 32 channels; derl/models.py:103,117-124)
// on the bf16 matrix cores WITHOUT giving up fp32 accuracy:
//   * a uint8 pixel is an integer < 256: exact in bf16 (8 significand bits);
//   * an fp32 operand w splits exactly into three bf16 terms w = hi + mid + lo (igemm_b3.hip);
//   * every bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32,
// so  sum_k x_k * w_k  costs THREE bf16 MFMAs of 32 cycles per 16 k instead of eight fp32 MFMAs of
// 64 cycles (5.3x less matrix time), with only the fp32 accumulation roundings any fp32 chain has.
// The 1/255 of the reference's `observations / 255` (models.py:121-123) is one division (div255) of the
// finished sum (forward) or of the slab element (weight gradient) instead of one rounding per term.
//
//   forward : A = 8 channel-bytes of two adjacent input pixels per lane (one ds_read_b64 from the
//             staged uint8 patch, converted in registers), B = three bf16 planes of the packed
//             weights in LDS (528-byte rows: conflict-free ds_read_b128).
//   wgrad   : dW[oc][k] = sum_m dY[m][oc] * x[m][k]; A = dY^T as three bf16 planes [oc][m] built once
//             per 256-pixel tile, B = the bytes of 8 pixels at this lane's k (8 ds_read_u8).
// Tiling, patch staging and the persistent tile loop are those of conv0.hip.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "conv0_tile.hpp"

namespace dx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo_f32(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f32(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// exact split of two floats into packed (hi, mid, lo) bf16 pairs
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = pack2(x0, x1);
  const float r0 = x0 - lo_f32(h), r1 = x1 - hi_f32(h);
  m = pack2(r0, r1);
  l = pack2(r0 - lo_f32(m), r1 - hi_f32(m));
}

// two bytes -> two bf16 (exact: the fp32 of an integer < 256 has a zero low half)
__device__ __forceinline__ uint32_t bytes_to_bf16x2(float f0, float f1) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, f1), __builtin_bit_cast(uint32_t, f0), 0x07060302u);
}

__device__ __forceinline__ bf16x8 expand8(uint2 w) {
  u32x4 r;
  r.x = bytes_to_bf16x2(static_cast<float>(w.x & 0xff), static_cast<float>((w.x >> 8) & 0xff));
  r.y = bytes_to_bf16x2(static_cast<float>((w.x >> 16) & 0xff), static_cast<float>(w.x >> 24));
  r.z = bytes_to_bf16x2(static_cast<float>(w.y & 0xff), static_cast<float>((w.y >> 8) & 0xff));
  r.w = bytes_to_bf16x2(static_cast<float>((w.y >> 16) & 0xff), static_cast<float>(w.y >> 24));
  return __builtin_bit_cast(bf16x8, r);
}

// x / 255 to within the last bit: multiply by the rounded reciprocal, one Newton residual step
// (3 instructions instead of the ~10 of an IEEE division; 32 of them per lane and tile)
__device__ __forceinline__ float div255(float x) {
  const float r = 1.0f / 255.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

constexpr int kWRowB = 528;                 // bytes per LDS weight row: 256 bf16 + 16 pad
constexpr int kWPlaneB = 32 * kWRowB;       // one plane of the packed weights

// T2 = 32-pixel MFMA tiles per wave: 2 -> four waves of 64 pixels, 1 -> eight waves of 32 pixels
// per 256-pixel tile.  Eight waves read every weight fragment twice as often (LDS 149 of 128
// B/clk/CU at full MFMA rate, i.e. a bound at 0.86) but put four waves on a SIMD: the phases of a
// tile that leave the matrix pipe idle (patch staging, prefetch issue, epilogue: 46 % of a tile
// with four waves, in-kernel stamps) are covered by other waves' MFMA loops, and a rollout-sized
// launch (one tile per workgroup) halves its dependent MFMA chain.
// Wb = the pre-split planes of dx_cnn_pack ([3][32][256] bf16; NULL: split Wp here).
template <int T2>
__global__ __launch_bounds__(512 / T2) void conv0_fwd_b16_kernel(const Conv0Args a, const uint16_t *Wb,
                                                                  unsigned long long *stamps) {
  constexpr int NT = 512 / T2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // DX_DIAG only: shader cycles per phase, summed over this workgroup's tiles (wave 0)
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
  const unsigned long long t_entry = (kDiag && stamps) ? __builtin_amdgcn_s_memtime() : 0;
#define DX_C0_MARK(i) if (kDiag && stamps) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph[i] += now - tprev; tprev = now; }
  uint8_t *Wpl = smem;                      // planes hi, mid, lo
  uint8_t *patch = smem + 3 * kWPlaneB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowB = a.in_w * 4;
  // the first patch is the long pole (HBM): its loads go out before the weights'
  u32x4 pre[kPatchRegs * 256 / NT];
  Seg nxt = tile_segments(a, min(static_cast<int>(blockIdx.x), a.ntiles - 1) * kTile);  // the tile whose patch is in `pre`
  if (blockIdx.x < a.ntiles) patch_load<NT>(a, nxt, pre);
  // the gather entries of the tile after that one: in flight for a whole tile before they are used
  TileImages ahead = tile_images(a, min(static_cast<int>(blockIdx.x + gridDim.x), a.ntiles - 1) * kTile);
  if (Wb) {  // 3 planes x 32 rows x 32 pieces of 16 B, every load in flight before the first LDS write
    u32x4 wv[3072 / NT];
#pragma unroll
    for (int u = 0; u < 3072 / NT; ++u) {
      const int i = u * NT + tid;
      wv[u] = *reinterpret_cast<const u32x4 *>(Wb + (i >> 10) * 8192 + ((i >> 5) & 31) * 256 + (i & 31) * 8);
    }
#pragma unroll
    for (int u = 0; u < 3072 / NT; ++u) {
      const int i = u * NT + tid;
      *reinterpret_cast<u32x4 *>(Wpl + (i >> 10) * kWPlaneB + ((i >> 5) & 31) * kWRowB + (i & 31) * 16) = wv[u];
    }
  } else {
    for (int i = tid; i < 32 * 64; i += NT) {  // packed weights [32][256] fp32 -> three bf16 planes
      const int row = i >> 6, c4 = i & 63;
      const float4 w = *reinterpret_cast<const float4 *>(a.Wp + row * 256 + 4 * c4);
      uint2 h, m, l;
      split2(w.x, w.y, h.x, m.x, l.x);
      split2(w.z, w.w, h.y, m.y, l.y);
      const int o = row * kWRowB + c4 * 8;
      *reinterpret_cast<uint2 *>(Wpl + o) = h;
      *reinterpret_cast<uint2 *>(Wpl + kWPlaneB + o) = m;
      *reinterpret_cast<uint2 *>(Wpl + 2 * kWPlaneB + o) = l;
    }
  }
  const int lrow = lane & 31, h = lane >> 5;
  const float bias = a.bias[lrow];
  // Whole tiles in the loop, the (at most one) ragged last tile after it: with the guarded stores of
  // a ragged tile inside the loop the number of stores per pass is not static, hipcc then waits
  // vmcnt(0) for the prefetched patch at the top of the next pass -- i.e. for every store of this
  // one.
  auto one_tile = [&](int tile, auto ragged) {
    constexpr bool RAGGED = decltype(ragged)::value;
    const int m0 = tile * kTile;
    const Seg s = nxt;
    if (kDiag && stamps) tprev = __builtin_amdgcn_s_memtime();
    __syncthreads();  // every wave finished reading the previous patch (and the planes are written)
    DX_C0_MARK(0)
    patch_store<NT>(s, pre, patch);
    DX_C0_MARK(1)
    // the next tile's loads go out as soon as their registers are free: before the barrier, so that
    // they are in flight while this wave waits for the others
    if (tile + gridDim.x < a.ntiles) {
      nxt = tile_segments(a, (tile + gridDim.x) * kTile, ahead);
      patch_load<NT>(a, nxt, pre);
      ahead = tile_images(a, min(static_cast<int>(tile + 2 * gridDim.x), a.ntiles - 1) * kTile);
    }
    __syncthreads();
    DX_C0_MARK(2)
    int rb[T2];
#pragma unroll
    for (int t2 = 0; t2 < T2; ++t2) rb[t2] = pixel_base(a, s, wave * (32 * T2) + t2 * 32 + lrow) + 8 * h;
    f32x16 acc[T2];
#pragma unroll
    for (int t2 = 0; t2 < T2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
    const uint8_t *wl = Wpl + lrow * kWRowB + 16 * h;
#pragma unroll 2
    for (int kh = 0; kh < 8; ++kh) {
      const uint8_t *prow = patch + kh * rowB;
#pragma unroll
      for (int c16 = 0; c16 < 2; ++c16) {
        // 16 k = (kh, kw = 4*c16 .. 4*c16+3, c): lane half h takes kw = 4*c16 + 2h, +1 (8 bytes)
        const int wo = (kh * 2 + c16) * 32;
        const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(wl + wo);
        const bf16x8 bm = *reinterpret_cast<const bf16x8 *>(wl + kWPlaneB + wo);
        const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(wl + 2 * kWPlaneB + wo);
#pragma unroll
        for (int t2 = 0; t2 < T2; ++t2) {
          const bf16x8 af = expand8(*reinterpret_cast<const uint2 *>(prow + rb[t2] + 16 * c16));
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bl, acc[t2], 0, 0, 0);
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bm, acc[t2], 0, 0, 0);
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bh, acc[t2], 0, 0, 0);
        }
      }
    }
    DX_C0_MARK(3)
    // /255, bias, ReLU, NHWC store: col = lane&31 (oc), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Full tiles (all but the last) store through one per-lane base pointer with immediate
    // offsets; a guarded store per element compiles to 32 exec-mask branches.
    float *obase = a.out + (static_cast<long long>(m0) + wave * (32 * T2) + 4 * h) * 32 + lrow;
    if (!RAGGED) {
#pragma unroll
      for (int t2 = 0; t2 < T2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = div255(acc[t2][r]) + bias;
          obase[(t2 * 32 + (r & 3) + 8 * (r >> 2)) * 32] = v > 0.f ? v : 0.f;
        }
    } else {
#pragma unroll
      for (int t2 = 0; t2 < T2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wave * (32 * T2) + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < a.M) {
            const float v = div255(acc[t2][r]) + bias;
            obase[(t2 * 32 + (r & 3) + 8 * (r >> 2)) * 32] = v > 0.f ? v : 0.f;
          }
        }
    }
    DX_C0_MARK(4)
  };
  const int nfull = a.M / kTile;
  int tile = blockIdx.x;
  const unsigned long long t_first = (kDiag && stamps) ? __builtin_amdgcn_s_memtime() : 0;
  for (; tile < nfull; tile += gridDim.x) one_tile(tile, std::false_type{});
  if (tile < a.ntiles) one_tile(tile, std::true_type{});
  if (kDiag && stamps && tid == 0) {
    unsigned long long *o = stamps + static_cast<long long>(blockIdx.x) * 8;
    for (int i = 0; i < 5; ++i) o[i] = ph[i];
    o[5] = t_first - t_entry;
    o[6] = __builtin_amdgcn_s_memtime() - t_entry;
    o[7] = __builtin_amdgcn_s_memrealtime();
  }
#undef DX_C0_MARK
}

constexpr int kGRowB = 528;            // bytes per row of a dY^T plane: 256 bf16 (m) + 16 pad
constexpr int kGPlaneB = 32 * kGRowB;

// GROUP4 (output width and pixels per image multiples of 4, i.e. every aligned group of four
// tile pixels lies in ONE output row): the patch offset of pixel 4g + e is base[g] + 16 e.  A lane's
// 16 group bases of a tile are read once before the MFMA loop, so the loop's LDS reads carry no
// dependent table lookup and the next 16 pixels' operands are in flight under the current 12 MFMAs.
// (Per-pixel table, read inside the loop: 11,100 cycles per tile for 6,144 of MFMA in round 2.)
template <bool GROUP4>
__global__ __launch_bounds__(256, 2) void conv0_wgrad_b16_kernel(const Conv0Args a, unsigned long long *stamps) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // DX_DIAG only: shader cycles per phase, summed over this workgroup's tiles (wave 0)
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
  const unsigned long long t_entry = (kDiag && stamps) ? __builtin_amdgcn_s_memtime() : 0;
#define DX_C0_MARK(i) if (kDiag && stamps) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph[i] += now - tprev; tprev = now; }
  uint8_t *Gpl = smem;                                             // dY^T planes hi, mid, lo: [oc][m]
  int *rbtab = reinterpret_cast<int *>(smem + 3 * kGPlaneB);       // [256] pixel (or [64] group) bases
  float *red = reinterpret_cast<float *>(smem + 3 * kGPlaneB + kTile * 4);  // [8][32] bias partials
  uint8_t *patch = smem + 3 * kGPlaneB + kTile * 4 + 8 * 32 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowB = a.in_w * 4;
  const int lcol = lane & 31, h = lane >> 5;
  // Wave roles: kgrp = wave & 1 owns kernel rows 4*kgrp .. 4*kgrp+3 (128 k), mgrp = wave >> 1 owns
  // pixels 128*mgrp .. +127 of every tile.  A lane reads one DWORD of an input row per pixel --
  // lane j: row 4*kgrp + j/8, bytes 4*(j%8) .. +3 -- and byte t of it is this lane's operand of k
  // tile t (k = 32*row + 4*(j%8) + t): one LDS read feeds four k tiles x three planes = 12 MFMAs
  // (a byte per read fed 3), which is what this kernel was bound by.
  const int kgrp = wave & 1, mgrp = wave >> 1;
  const int koff = (4 * kgrp + (lcol >> 3)) * rowB + 4 * (lcol & 7);
  // staging role: thread -> (oc = tid & 31, 32 pixels mg*32 .. mg*32+31)
  const int soc = tid & 31, mg = tid >> 5;
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bias_acc = 0.f;
  u32x4 pre[kPatchRegs];
  float gpre[32];
  // dY0[m][oc]: 32 lanes = 128 contiguous bytes per pixel.  Whole tiles: one base pointer and
  // immediate offsets; the ragged last tile clamps its rows (the clamped values are masked below)
  auto g_load = [&](int m0) {
    const int nvalid = min(kTile, a.M - m0);
    const float *gp = a.G + static_cast<long long>(m0 + mg * 32) * 32 + soc;
    if (nvalid == kTile) {
#pragma unroll
      for (int u = 0; u < 32; ++u) gpre[u] = gp[u * 32];
    } else {
#pragma unroll
      for (int u = 0; u < 32; ++u) gpre[u] = a.G[static_cast<long long>(m0 + min(mg * 32 + u, nvalid - 1)) * 32 + soc];
    }
  };
  Seg nxt = tile_segments(a, min(static_cast<int>(blockIdx.x), a.ntiles - 1) * kTile);  // the tile whose patch is in `pre`
  if (blockIdx.x < a.ntiles) {
    patch_load(a, nxt, pre);
    g_load(blockIdx.x * kTile);
  }
  // the gather entries of the tile after that one: in flight for a whole tile before they are used
  TileImages ahead = tile_images(a, min(static_cast<int>(blockIdx.x + gridDim.x), a.ntiles - 1) * kTile);
  const unsigned long long t_first = (kDiag && stamps) ? __builtin_amdgcn_s_memtime() : 0;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int m0 = tile * kTile;
    const Seg s = nxt;
    if (kDiag && stamps) tprev = __builtin_amdgcn_s_memtime();
    __syncthreads();
    DX_C0_MARK(0)
    patch_store(s, pre, patch);
    {
      const int nvalid = min(kTile, a.M - m0);
      const bool whole = nvalid == kTile;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // 8 pixels -> one 16-byte store per plane
        u32x4 ph4, pm4, pl4;
        uint32_t hh[4], mm[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = mg * 32 + q * 8 + 2 * e;
          const float g0 = (whole || m < nvalid) ? gpre[q * 8 + 2 * e] : 0.f;
          const float g1 = (whole || m + 1 < nvalid) ? gpre[q * 8 + 2 * e + 1] : 0.f;
          bias_acc += g0 + g1;
          split2(g0, g1, hh[e], mm[e], ll[e]);
        }
        ph4.x = hh[0]; ph4.y = hh[1]; ph4.z = hh[2]; ph4.w = hh[3];
        pm4.x = mm[0]; pm4.y = mm[1]; pm4.z = mm[2]; pm4.w = mm[3];
        pl4.x = ll[0]; pl4.y = ll[1]; pl4.z = ll[2]; pl4.w = ll[3];
        const int o = soc * kGRowB + (mg * 32 + q * 8) * 2;
        *reinterpret_cast<u32x4 *>(Gpl + o) = ph4;
        *reinterpret_cast<u32x4 *>(Gpl + kGPlaneB + o) = pm4;
        *reinterpret_cast<u32x4 *>(Gpl + 2 * kGPlaneB + o) = pl4;
      }
    }
    if (GROUP4) {
      if (tid < kTile / 4) rbtab[tid] = pixel_base(a, s, 4 * tid);
    } else {
      rbtab[tid] = pixel_base(a, s, tid);
    }
    DX_C0_MARK(1)
    // the next tile's loads go out as soon as their registers are free: before the barrier, so that
    // they are in flight while this wave waits for the others
    if (tile + gridDim.x < a.ntiles) {
      nxt = tile_segments(a, (tile + gridDim.x) * kTile, ahead);
      patch_load(a, nxt, pre);
      g_load((tile + gridDim.x) * kTile);
      ahead = tile_images(a, min(static_cast<int>(tile + 2 * gridDim.x), a.ntiles - 1) * kTile);
    }
    __syncthreads();
    DX_C0_MARK(2)
    const uint8_t *gl = Gpl + lcol * kGRowB + 16 * h;
    // 12 MFMAs on the operands of 16 pixels: lane half h has pixels 16c + 8h .. +7 (dword w[e]
    // of pixel e: byte t = this lane's k of k tile t)
    auto mfma16 = [&](const uint32_t (&w)[8], bf16x8 ah, bf16x8 am, bf16x8 al) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = static_cast<float>((w[e] >> (8 * t)) & 0xff);
        u32x4 b;
        b.x = bytes_to_bf16x2(x[0], x[1]); b.y = bytes_to_bf16x2(x[2], x[3]);
        b.z = bytes_to_bf16x2(x[4], x[5]); b.w = bytes_to_bf16x2(x[6], x[7]);
        const bf16x8 bf = __builtin_bit_cast(bf16x8, b);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bf, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf, acc[t], 0, 0, 0);
      }
    };
    if constexpr (GROUP4) {
      int gb[kTile / 32][2];  // patch offsets (+ this lane's k offset) of its two pixel groups per step
#pragma unroll
      for (int cc = 0; cc < kTile / 32; ++cc) {
        const int2 g = *reinterpret_cast<const int2 *>(&rbtab[4 * (mgrp * (kTile / 32) + cc) + 2 * h]);
        gb[cc][0] = g.x + koff;
        gb[cc][1] = g.y + koff;
      }
      uint32_t wc[8];
      bf16x8 ahc, amc, alc;
      auto frag_load = [&](int cc, uint32_t (&w)[8], bf16x8 &ah, bf16x8 &am, bf16x8 &al) {
        const int c = mgrp * (kTile / 32) + cc;
        ah = *reinterpret_cast<const bf16x8 *>(gl + c * 32);
        am = *reinterpret_cast<const bf16x8 *>(gl + kGPlaneB + c * 32);
        al = *reinterpret_cast<const bf16x8 *>(gl + 2 * kGPlaneB + c * 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = *reinterpret_cast<const uint32_t *>(patch + gb[cc][e >> 2] + 16 * (e & 3));
      };
      frag_load(0, wc, ahc, amc, alc);
#pragma unroll
      for (int cc = 0; cc < kTile / 32; ++cc) {
        uint32_t wn[8];
        bf16x8 ahn, amn, aln;
        if (cc + 1 < kTile / 32) frag_load(cc + 1, wn, ahn, amn, aln);
        __builtin_amdgcn_sched_barrier(0);  // the next step's reads are issued before this step's MFMAs
        mfma16(wc, ahc, amc, alc);
        if (cc + 1 < kTile / 32) {
#pragma unroll
          for (int e = 0; e < 8; ++e) wc[e] = wn[e];
          ahc = ahn; amc = amn; alc = aln;
        }
      }
    } else {
#pragma unroll 2
      for (int cc = 0; cc < kTile / 32; ++cc) {  // 16 pixels per MFMA: lane half h takes pixels 16c + 8h .. +7
        const int c = mgrp * (kTile / 32) + cc;
        const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(gl + c * 32);
        const bf16x8 am = *reinterpret_cast<const bf16x8 *>(gl + kGPlaneB + c * 32);
        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(gl + 2 * kGPlaneB + c * 32);
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = *reinterpret_cast<const uint32_t *>(patch + rbtab[16 * c + 8 * h + e] + koff);
        mfma16(w, ah, am, al);
      }
    }
    DX_C0_MARK(3)
  }
  // The two pixel halves meet in LDS (the dY planes are dead now), then
  // slab[block][oc][k]: rows = oc, cols = k; the input scale 1/255 is applied here
  __syncthreads();
  float *part = reinterpret_cast<float *>(smem);  // [kgrp][t][r][lane]
  if (mgrp == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[((kgrp * 4 + t) * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (mgrp == 0) {
    float *slab = a.slab + static_cast<long long>(blockIdx.x) * 32 * 256;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = (4 * kgrp + (lcol >> 3)) * 32 + 4 * (lcol & 7) + t;
        slab[oc * 256 + k] = div255(acc[t][r] + part[((kgrp * 4 + t) * 16 + r) * 64 + lane]);
      }
  }
  if (a.bias_slab) {
    __syncthreads();
    red[mg * 32 + soc] = bias_acc;
    __syncthreads();
    if (tid < 32) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += red[g * 32 + tid];
      a.bias_slab[static_cast<long long>(blockIdx.x) * 32 + tid] = v;
    }
  }
  if (kDiag && stamps && tid == 0) {
    unsigned long long *o = stamps + static_cast<long long>(blockIdx.x) * 8;
    for (int i = 0; i < 5; ++i) o[i] = ph[i];
    o[5] = t_first - t_entry;
    o[6] = __builtin_amdgcn_s_memtime() - t_entry;
    o[7] = __builtin_amdgcn_s_memrealtime();
  }
#undef DX_C0_MARK
}

// Rollout batches (a few thousand output pixels): one 32-pixel x 32-channel tile per workgroup,
// the four waves split the 16 K chunks (kernel rows 2w, 2w+1), every operand goes straight from
// global memory into the MFMA layout -- 8 input bytes and 3 x 16 bytes of the pre-split weight
// planes (dx_cnn_pack) per lane and chunk, all 16 loads in flight before the first of 12 MFMAs --
// and the four partial tiles meet in LDS once (igemm_lat.hip has the fp32 layers of this path).
__global__ __launch_bounds__(256) void conv0_lat_b16_kernel(const Conv0Args a, const uint16_t *Wb) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int rowB = a.in_w * 4;
  const int P = a.h0 * a.w0;
  const int m = min(static_cast<int>(blockIdx.x) * 32 + r, a.M - 1);  // clamped rows are not stored
  int img = m / P;
  const int pix = m - img * P;
  const int oy = pix / a.w0, ox = pix - oy * a.w0;
  if (a.idx) img = a.idx[img];
  const uint8_t *src = a.obs + static_cast<long long>(img) * a.in_h * rowB + (4 * oy + 2 * wave) * rowB + ox * 16 + 8 * h;
  const uint16_t *wsrc = Wb + r * 256 + (2 * wave) * 32 + 8 * h;
  uint2 araw[4];
  u32x4 braw[3][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // chunk c: kernel row 2*wave + (c >> 1), bytes 16*(c & 1) .. +15
    araw[c] = *reinterpret_cast<const uint2 *>(src + (c >> 1) * rowB + 16 * (c & 1));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      braw[pl][c] = *reinterpret_cast<const u32x4 *>(wsrc + pl * 8192 + (c >> 1) * 32 + 16 * (c & 1));
  }
  __builtin_amdgcn_sched_barrier(0);  // all loads before the first MFMA (see igemm_lat.hip)
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bf16x8 af = expand8(araw[c]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, braw[2][c]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, braw[1][c]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, braw[0][c]), acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
  __syncthreads();
  const float bias = a.bias[r];
#pragma unroll
  for (int t = 0; t < 4; ++t) {  // wave w finishes accumulator registers 4w .. 4w+3
    const int i = 4 * wave + t;
    const float sum = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
    const int mm = blockIdx.x * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (mm < a.M) {
      const float v = div255(sum) + bias;
      a.out[static_cast<long long>(mm) * 32 + r] = v > 0.f ? v : 0.f;
    }
  }
}

}  // namespace

namespace {
// DX_C0_WAVES=4|8 pins the waves per workgroup of the 256-pixel-tile forward (A/B runs)
int conv0_waves_env() {
  const int v = DX_ENV("DX_C0_WAVES", 0);
  return v;
}

template <int T2>
int launch_conv0_fwd_b16_as(const Conv0Args &a, const uint16_t *Wb, int lds, hipStream_t stream) {
  DX_LDS_OPT_IN(conv0_fwd_b16_kernel<T2>, 160 * 1024);
  int grid = a.ntiles < 512 ? a.ntiles : 512;
#if DX_DIAG
  if (getenv("DX_C0_DIAG")) {  // in-kernel phase cycles, summarised on stderr (synchronous)
    unsigned long long *dev = nullptr;
    DX_HIP(hipMalloc(&dev, static_cast<size_t>(grid) * 64));
    hipLaunchKernelGGL(conv0_fwd_b16_kernel<T2>, dim3(grid), dim3(512 / T2), lds, stream, a, Wb, dev);
    DX_LAUNCH_CHECK();
    DX_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(static_cast<size_t>(grid) * 8);
    DX_HIP(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    DX_HIP(hipFree(dev));
    double sum[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < grid; ++b)
      for (int i = 0; i < 7; ++i) sum[i] += static_cast<double>(h[static_cast<size_t>(b) * 8 + i]);
    const double tiles = static_cast<double>(a.ntiles);
    fprintf(stderr, "[conv0_fwd_b16<%d waves> M=%d tiles=%d grid=%d] cycles per tile (wave 0): barrier-in %.0f, patch store "
            "%.0f, prefetch issue + barrier %.0f, mfma loop %.0f, epilogue %.0f | prologue %.0f, whole kernel %.0f cycles per "
            "workgroup\n", 8 / T2, a.M, a.ntiles, grid, sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles,
            sum[4] / tiles, sum[5] / grid, sum[6] / grid);
    return DX_OK;
  }
#endif
  hipLaunchKernelGGL(conv0_fwd_b16_kernel<T2>, dim3(grid), dim3(512 / T2), lds, stream, a, Wb,
                     static_cast<unsigned long long *>(nullptr));
  DX_LAUNCH_CHECK();
  return DX_OK;
}
}  // namespace

// Wb: the pre-split weight planes of dx_cnn_pack, or NULL (the kernel then splits a.Wp itself)
int launch_conv0_fwd_b16(const Conv0Args &a, const uint16_t *Wb, hipStream_t stream) {
  DX_REQUIRE(a.obs && a.Wp && a.bias && a.out && a.M > 0, "conv0_fwd_b16: bad arguments");
  const int lds = 3 * kWPlaneB + patch_bytes(a);
  DX_REQUIRE(lds <= 160 * 1024 && patch_bytes(a) <= kPatchRegs * 256 * 16,
             "conv0_fwd_b16: tile does not fit (%d LDS bytes, patch %d)", lds, patch_bytes(a));
  // eight waves up to ~5 tiles per resident workgroup slot (measured: 128 images 11.6 -> 9.8 us, 256: 15.1 -> 13.6,
  // 1024: 32.9 -> 31.3); at 8192 images the launch sits at the package power limit either way (167 vs 170 us)
  const int waves = conv0_waves_env() ? conv0_waves_env() : (a.ntiles <= 2560 ? 8 : 4);
  return waves == 8 ? launch_conv0_fwd_b16_as<1>(a, Wb, lds, stream) : launch_conv0_fwd_b16_as<2>(a, Wb, lds, stream);
}

// DX_ENOSUP when the shape is not covered (odd row pitch) or the batch is too big for this path
int launch_conv0_lat_b16(const Conv0Args &a, const uint16_t *Wb, hipStream_t stream) {
  DX_REQUIRE(a.obs && Wb && a.bias && a.out && a.M > 0, "conv0_lat_b16: bad arguments");
  if (a.in_w % 2 || (static_cast<long long>(a.in_h) * a.in_w) % 2 || a.w0 < 1) return DX_ENOSUP;
  hipLaunchKernelGGL(conv0_lat_b16_kernel, dim3(cdiv(a.M, 32)), dim3(256), 0, stream, a, Wb);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

namespace {
template <bool GROUP4>
int launch_conv0_wgrad_b16_as(const Conv0Args &a, int nblocks, int lds, hipStream_t stream) {
  DX_LDS_OPT_IN(conv0_wgrad_b16_kernel<GROUP4>, 160 * 1024);
#if DX_DIAG
  if (getenv("DX_C0_DIAG")) {  // in-kernel phase cycles, summarised on stderr (synchronous)
    unsigned long long *dev = nullptr;
    DX_HIP(hipMalloc(&dev, static_cast<size_t>(nblocks) * 64));
    hipLaunchKernelGGL(conv0_wgrad_b16_kernel<GROUP4>, dim3(nblocks), dim3(256), lds, stream, a, dev);
    DX_LAUNCH_CHECK();
    DX_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(static_cast<size_t>(nblocks) * 8);
    DX_HIP(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    DX_HIP(hipFree(dev));
    double sum[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b)
      for (int i = 0; i < 7; ++i) sum[i] += static_cast<double>(h[static_cast<size_t>(b) * 8 + i]);
    const double tiles = static_cast<double>(a.ntiles);
    fprintf(stderr, "[conv0_wgrad_b16<group4=%d> M=%d tiles=%d grid=%d] cycles per tile (wave 0): barrier-in %.0f, staging "
            "%.0f, prefetch issue + barrier %.0f, mfma loop %.0f | prologue %.0f, whole kernel %.0f cycles per workgroup\n",
            GROUP4 ? 1 : 0, a.M, a.ntiles, nblocks, sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles,
            sum[5] / nblocks, sum[6] / nblocks);
    return DX_OK;
  }
#endif
  hipLaunchKernelGGL(conv0_wgrad_b16_kernel<GROUP4>, dim3(nblocks), dim3(256), lds, stream, a,
                     static_cast<unsigned long long *>(nullptr));
  DX_LAUNCH_CHECK();
  return DX_OK;
}
}  // namespace

int launch_conv0_wgrad_b16(const Conv0Args &a, int nblocks, hipStream_t stream) {
  DX_REQUIRE(a.obs && a.G && a.slab && a.M > 0 && nblocks >= 1 && nblocks <= a.ntiles,
             "conv0_wgrad_b16: bad arguments");
  const int lds = 3 * kGPlaneB + kTile * 4 + 8 * 32 * 4 + patch_bytes(a);
  DX_REQUIRE(lds <= 160 * 1024 && patch_bytes(a) <= kPatchRegs * 256 * 16,
             "conv0_wgrad_b16: tile does not fit (%d LDS bytes, patch %d)", lds, patch_bytes(a));
  // DX_C0_GROUP4=0: the per-pixel offset table for every shape (A/B runs)
  const bool allow4 = DX_ENV("DX_C0_GROUP4", 1) != 0;
  const bool group4 = allow4 && a.w0 % 4 == 0 && (a.h0 * a.w0) % 4 == 0;
  return group4 ? launch_conv0_wgrad_b16_as<true>(a, nblocks, lds, stream)
                : launch_conv0_wgrad_b16_as<false>(a, nblocks, lds, stream);
}

}  // namespace dx

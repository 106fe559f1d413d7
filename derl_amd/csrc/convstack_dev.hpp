// Device helpers shared by the image-resident conv-stack kernels (convstack.hip: the rollout step / act;
// convstack_train.hip: the forward of a training minibatch): LDS layout of the bf16 plane images, conv0 on
// v_mfma_f32_32x32x16_bf16 from the raw frame bytes, conv1 / conv2 K steps on v_mfma_f32_16x16x32_bf16 with both
// operands split exactly into three bf16 terms.  derl/models.py:104-111.
#pragma once
#include "bf16_split.hpp"
#include "igemm.hpp"

namespace dx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kTerms = 6;  // products per fp32 x fp32 (6: everything above 2^-23 of the product; 9: all)
constexpr int kIn = 84, kFrameB = kIn * kIn * 4, kRowB = kIn * 4;  // uint8 NHWC frame, 4 stacked channels
constexpr int kP0 = 400, kP1 = 81, kP2 = 49;                       // output pixels of the three layers
// LDS images of the activations: pixel pitch and ROW pitch chosen (exhaustive search over paddings) so that
// every ds_read_b128 of a B fragment -- 16 consecutive output pixels of a 9- / 7-wide image, i.e. with a
// row wrap inside the tile, x 4 k groups -- puts its four 16-lane groups on 16 distinct 16-byte bank
// units: unpadded rows gave 2- and 3-way conflicts on a third of the reads (conv1 1.8x, conv2 2.5x the
// LDS cycles, and the LDS pipe is what these loops lean on)
constexpr int kY0P = 80, kY0R = 20 * kY0P + 16, kY0Plane = 20 * kY0R;    // bytes per y0 pixel / row / plane (32 bf16 + pad)
constexpr int kY1P = 160, kY1R = 9 * kY1P + 192, kY1Plane = 9 * kY1R;    // bytes per y1 pixel / row / plane (64 bf16 + pad)
constexpr int kWRowB = 528, kWPlaneB = 32 * kWRowB;                // conv0 weight planes in LDS: 256 bf16 + 16 B pad
// LDS: [conv0 weight planes][region B][tail sums].  Region B holds, in turn: the frame while conv0 multiplies, the three
// y0 planes (from its start), then the three y1 planes (at its start) beside the NEXT image's frame (behind them).  The
// frame lies there as BF16 (a uint8 pixel is exact in bf16): every byte is converted ONCE, when the frame is written,
// instead of once per use -- each byte is an operand of four 8 x 8 patches, and the conversion (1.5 vector-ALU
// instructions per byte and use) was the bulk of the conv0 loops' instructions (one wave per SIMD ran four tiles in
// 16,300 cycles for 4,600 of matrix time).
constexpr int kFrame16B = 2 * kFrameB;  // 84 x 84 x 4 bf16
constexpr int oW0 = 0, oB = oW0 + 3 * kWPlaneB, oY0 = oB, oY1 = oB, oFrame = oB + 3 * kY1Plane;
constexpr int kRegionB = 3 * kY0Plane > 3 * kY1Plane + kFrame16B ? 3 * kY0Plane : 3 * kY1Plane + kFrame16B;
constexpr int oTail = oB + kRegionB, kTailOut = 24, kLdsBytes = oTail + 2 * 8 * kTailOut * 4;  // (tail: [step parity][8 waves][up to 24 padded outputs])
// conv0's 13th tile (pixels 384 .. 399) is multiplied in two K halves by the first two B waves: the second half's partial
// sums cross to wave 0 through this area (one f32x16 per lane), behind the barrier that ends the phase
constexpr int oExch = kLdsBytes, kExchB = 64 * 64, oBias0 = oExch + kExchB, kLdsBytesX = oBias0 + 32 * 4;  // + conv0's bias (read per image)
static_assert(oB % 16 == 0 && oFrame % 16 == 0 && oTail % 16 == 0 && oExch % 16 == 0 && kLdsBytesX <= 160 * 1024, "LDS layout");
// conv0's 32 biases live in LDS for the launch: the epilogue of every image reads this lane's 16 of them (four
// ds_read_b128 behind the phase's barrier; as global loads they were an exposed L2 round trip per image)
__device__ __forceinline__ void bias0_to_lds(uint8_t *smem, const float *bias0, int tid) {
  if (tid < 32) *reinterpret_cast<float *>(smem + oBias0 + 4 * tid) = bias0[tid];
}
__device__ __forceinline__ void bias0_from_lds(const uint8_t *smem, int lane, f32x4 (&b)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const f32x4 *>(smem + oBias0 + 4 * (8 * q + 4 * (lane >> 5)));
}
__device__ __forceinline__ void exch_put(uint8_t *smem, int lane, const f32x16 &v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(smem + oExch + q * 1024 + lane * 16) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
__device__ __forceinline__ void exch_add(const uint8_t *smem, int lane, f32x16 &v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 w = *reinterpret_cast<const f32x4 *>(smem + oExch + q * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * q + j] += w[j];
  }
}

// two bytes -> two bf16 (exact: the fp32 of an integer < 256 has a zero low half)
__device__ __forceinline__ uint32_t cs_bytes_to_bf16x2(float f0, float f1) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, f1), __builtin_bit_cast(uint32_t, f0), 0x07060302u);
}
__device__ __forceinline__ bf16x8 cs_expand8(uint2 w) {
  u32x4 r;
  r.x = cs_bytes_to_bf16x2(static_cast<float>(w.x & 0xff), static_cast<float>((w.x >> 8) & 0xff));
  r.y = cs_bytes_to_bf16x2(static_cast<float>((w.x >> 16) & 0xff), static_cast<float>(w.x >> 24));
  r.z = cs_bytes_to_bf16x2(static_cast<float>(w.y & 0xff), static_cast<float>((w.y >> 8) & 0xff));
  r.w = cs_bytes_to_bf16x2(static_cast<float>((w.y >> 16) & 0xff), static_cast<float>(w.y >> 24));
  return __builtin_bit_cast(bf16x8, r);
}
// x / 255 to within the last bit (conv0_b16.hip: div255)
__device__ __forceinline__ float cs_div255(float x) {
  const float r = 1.0f / 255.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// Workgroup barrier for LDS hand-offs ONLY: waits for this wave's LDS operations, not for its global
// loads.  __syncthreads() is a fence and drains vmcnt too -- here that would park every wave at each
// barrier until the NEXT layer's weight fragments (24-27 KB per wave, issued on purpose a layer ahead)
// have landed: 12,000 of the first version's 54,000 cycles.  Nothing crosses these barriers through
// global memory.
// A per-iteration copy of a loop-invariant value the compiler cannot see through: what is derived from it
// is recomputed inside the step loop (a few integer instructions) instead of being hoisted out of the loop
// and kept in registers across all of it -- with the hoisted per-lane tables of three layers alive the
// kernel needed 800 bytes of scratch per lane.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// 16 bytes from a UNIFORM base + a per-lane 32-bit byte offset: the scalar-base form of global_load (one
// offset register per lane, the plane / step offsets go into the scalar base and the immediate) instead
// of one 64-bit per-lane address per fragment
// (the pointer is cast to the GLOBAL address space explicitly: behind opaque_base the compiler no longer knows where it
// came from and would emit flat loads -- per-lane 64-bit addresses, and counted in lgkmcnt as well as vmcnt, which
// every LDS wait of these kernels would then trip over)
__device__ __forceinline__ u32x4 load16(const void *base_uniform, unsigned lane_bytes) {
  using gchar = __attribute__((address_space(1))) const char;
  using gvec = __attribute__((address_space(1))) const u32x4;
  return *reinterpret_cast<gvec *>((gchar *)(base_uniform) + lane_bytes);
}

// A uniform pointer the compiler cannot see through: every piece's 64-bit base is then formed where it is used (two
// scalar adds) instead of being hoisted out of the image loop -- hoisted, the ~90 piece bases of the two layers do not
// fit the scalar registers and came back through v_readlane pairs (vector-ALU slots, inside the MFMA loops)
template <class T>
__device__ __forceinline__ const T *opaque_base(const T *p) {
  asm volatile("" : "+s"(p));
  return p;
}

// One KB piece of a fragment-ordered weight copy: `base` (uniform) + `elems` bf16 + the lane's 16 bytes, as
// global_load_dwordx4 v, v_lane_offset, s[base'] -- base' formed where it is used (opaque_base), and the lane offset
// made opaque in the same block so that its 32 -> 64 bit extension is not hoisted into a register pair (the scalar-base
// form is only selected when the extension is visible next to the load)
__device__ __forceinline__ u32x4 load_piece(const uint16_t *base, int elems, unsigned lane_bytes) {
  // (opaque twice: the inner one keeps base + elems from being hoisted, the outer one keeps the constant out of the
  // per-lane part of the address, where it would be added on the vector ALU)
  return load16(opaque_base(opaque_base(base) + elems), static_cast<unsigned>(opaque(static_cast<int>(lane_bytes))));
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ReLU that keeps a NaN a NaN (`x > 0 ? x : 0` and v_max_f32 turn it into 0).  The exact split of an overflowed (Inf)
// activation is (Inf, NaN, NaN) -- an fp32 chain would have kept Inf -- so the next layer's sums over it are NaN: they
// must stay NaN down to the outputs instead of being zeroed one ReLU later (tests/test_cnn_gpu.py:
// test_exact_split_keeps_extreme_magnitudes_and_never_hides_a_non_finite_value)
__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.f ? 0.f : x; }

// four activations (consecutive channels of one pixel) -> 8 bytes in each of the three bf16 planes
__device__ __forceinline__ void store_planes4(uint8_t *smem, int off, int plane_bytes, f32x4 v) {
  const Split4 s = split4(v);
  *reinterpret_cast<uint2 *>(smem + off) = s.hi;
  *reinterpret_cast<uint2 *>(smem + off + plane_bytes) = s.mid;
  *reinterpret_cast<uint2 *>(smem + off + 2 * plane_bytes) = s.lo;
}

// The LDS frame: 84 x 84 pixels x 4 channels as bf16, in 16-byte chunks of two pixels.  The chunks of a row go to TWO
// planes by parity -- chunk 2 j of row y to plane 0, chunk 2 j + 1 to plane 1, both at index 21 y + j -- because conv0's
// operand read of lane (pixel, k half kg) is chunk 2 ox + 2 (c & 1) + kg of its row: with the chunks in row-major order the
// 16 lanes of an LDS cycle (one kg, 16 pixels) hit only the even or only the odd 16-byte bank quads, 7.8 cycles per
// ds_read_b128 for 4 (SQ_LDS_BANK_CONFLICT of the diag variants: 775 conflict cycles per image in these reads, 220 more
// in the stores below); per plane they are 16 consecutive chunks.
constexpr int kFramePlaneB = kFrame16B / 2;
static_assert(kFramePlaneB == 16 * 21 * kIn, "21 chunks of each parity per row");
// 16 raw bytes of a frame (16-byte unit `unit` of its 28,224 bytes = four pixels = one chunk of each parity) -> 16 bf16
// in the LDS frame: converted here, once
__device__ __forceinline__ void put_frame_unit(uint8_t *smem, int unit, u32x4 raw) {
  *reinterpret_cast<bf16x8 *>(smem + oFrame + 16 * unit) = cs_expand8(uint2{raw.x, raw.y});
  *reinterpret_cast<bf16x8 *>(smem + oFrame + kFramePlaneB + 16 * unit) = cs_expand8(uint2{raw.z, raw.w});
}

// conv0 for NT_ 32-pixel tiles (tile0, tile0 + TS, ...) of this wave: D[channel][pixel] = sum over the 16 K
// chunks of W0(planes lo, mid, hi) x pixels; every weight fragment is read from LDS once per chunk for all of the
// wave's tiles.  One scheduling region per chunk: its 3 NT_ MFMAs with the NEXT chunk's 3 + NT_ fragment reads
// interleaved, one read behind each of the first MFMAs (no vector-ALU work: the pixels are bf16 in LDS).
// (C0 .. C1 - 1: the K chunks this call multiplies -- all sixteen, or one half of them where two waves share a tile and add
// their partial sums afterwards)
template <int NT_, int TS = 8, int NA = 2, int C0 = 0, int C1 = 16>
__device__ __forceinline__ void conv0_mfma(const uint8_t *smem, int tile0, int lane, f32x16 (&acc)[NA]) {
  static_assert(NT_ <= NA && 0 <= C0 && C0 < C1 && C1 <= 16, "accumulator tiles; chunk range");
  const int r = lane & 31, kg = lane >> 5;
  int pb[NT_];
#pragma unroll
  for (int t = 0; t < NT_; ++t) {
    const int p = min(32 * (tile0 + TS * t) + r, kP0 - 1);  // columns past the image compute a copy that is not stored
    const int oy = p / 20, ox = p - 20 * oy;
    pb[t] = oFrame + kg * kFramePlaneB + 16 * (kIn * oy + ox);  // chunk 21 (4 oy) + ox of plane kg
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  }
  const int wb = oW0 + r * kWRowB + 16 * kg;
  bf16x8 px[NT_];
  u32x4 wf[3];
  constexpr int aoff0 = 16 * ((C0 >> 1) * 21 + (C0 & 1));
#pragma unroll
  for (int t = 0; t < NT_; ++t) px[t] = *reinterpret_cast<const bf16x8 *>(smem + pb[t] + aoff0);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const u32x4 *>(smem + wb + pl * kWPlaneB + 32 * C0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = C0; c < C1; ++c) {  // chunk c: kernel row c / 2, bytes 16 (c % 2) .. + 15 of its 32
    bf16x8 pxn[NT_];
    u32x4 wfn[3];
    if (c + 1 < C1) {
      const int aoff = 16 * (((c + 1) >> 1) * 21 + ((c + 1) & 1));  // kernel row (c + 1) / 2, pixels 4 ((c + 1) % 2) + 2 kg ..
#pragma unroll
      for (int t = 0; t < NT_; ++t) pxn[t] = *reinterpret_cast<const bf16x8 *>(smem + pb[t] + aoff);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wfn[pl] = *reinterpret_cast<const u32x4 *>(smem + wb + pl * kWPlaneB + 32 * (c + 1));
    }
#pragma unroll
    for (int pl = 2; pl >= 0; --pl)
#pragma unroll
      for (int t = 0; t < NT_; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wf[pl]), px[t], acc[t], 0, 0, 0);
    if (c + 1 < C1) {
      constexpr int kReads = 3 + NT_, kMfma = 3 * NT_;
#pragma unroll
      for (int i = 0; i < (kReads < kMfma ? kReads : kMfma); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (kReads > kMfma) __builtin_amdgcn_sched_group_barrier(0x100, kReads - kMfma, 0);
      if (kMfma > kReads) __builtin_amdgcn_sched_group_barrier(0x008, kMfma - kReads, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < C1) {
#pragma unroll
      for (int t = 0; t < NT_; ++t) px[t] = pxn[t];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wf[pl] = wfn[pl];
    }
  }
}

// bias + ReLU + exact three-way split -> the y0 planes.  C/D layout of a 32x32 tile: column (pixel) =
// lane & 31, rows (channels) of register i = (i & 3) + 8 (i >> 2) + 4 (lane >> 5): registers 4q .. 4q + 3
// are four consecutive channels
template <int NT_, int TS = 8, int NA = 2>
__device__ __forceinline__ void conv0_store(uint8_t *smem, int tile0, int lane, const f32x16 (&acc)[NA], const f32x4 (&bias)[4],
                                            float *gy0) {
  const int r = lane & 31, kg = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT_; ++t) {
    const int p = 32 * (tile0 + TS * t) + r;
    if (p >= kP0) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = cs_div255(acc[t][4 * q + j]) + bias[q][j];
        v[j] = relu_keep_nan(x);
      }
      store_planes4(smem, oY0 + (p / 20) * kY0R + (p % 20) * kY0P + (8 * q + 4 * kg) * 2, kY0Plane, v);
      if (gy0) *reinterpret_cast<f32x4 *>(gy0 + p * 32 + 8 * q + 4 * kg) = v;  // (training: kept for the backward)
    }
  }
}

// the products of one 16x16 tile and one K step: weight fragment planes w (hi, mid, lo) x activation
// fragment planes x, smallest terms first -- in two parts, so that the next tile's LDS reads can be pinned
// between the first product and the rest (left alone, hipcc sinks every read to just before its use and
// waits for it at once: 3 exposed LDS round trips per 6 MFMAs, the first build's conv1 at 38 % of the pipe)
#define DX_CS_T(a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w[a]), as_bf16x8(x[b]), acc, 0, 0, 0);
__device__ __forceinline__ f32x4 mac_first(f32x4 acc, const u32x4 (&w)[3], const u32x4 (&x)[3]) {
  if (kTerms == 9) { DX_CS_T(2, 2) } else { DX_CS_T(2, 0) }
  return acc;
}
__device__ __forceinline__ f32x4 mac_rest(f32x4 acc, const u32x4 (&w)[3], const u32x4 (&x)[3]) {
  if (kTerms == 9) { DX_CS_T(2, 1) DX_CS_T(1, 2) DX_CS_T(2, 0) }
  DX_CS_T(0, 2) DX_CS_T(1, 1) DX_CS_T(1, 0) DX_CS_T(0, 1) DX_CS_T(0, 0)
  return acc;
}
#undef DX_CS_T

// One K half of conv1 (KH: taps 8 KH .. 8 KH + 7) or conv2 (KH: steps 9 KH .. 9 KH + 8 of 18) for this wave's
// 16 channels: NT pixel tiles, activations from the LDS planes (pb = byte address of the lane's pixel
// and k group in plane 0), the next tile's fragments read before this tile's MFMAs.
template <int LAYER, int KH, int S>
__device__ __forceinline__ constexpr int step_offset() {
  if (LAYER == 1) return ((8 * KH + S) >> 2) * kY0R + (S & 3) * kY0P;  // tap (kh, kw) = ((8 KH + s) / 4, s % 4): 32 input channels = one K step
  return (((9 * KH + S) >> 1) / 3) * kY1R + (((9 * KH + S) >> 1) % 3) * kY1P + ((9 * KH + S) & 1) * 64;  // step g = 9 KH + s: tap g / 2, channels 32 (g % 2) ..
}

// the activation fragments (three planes) of tile pair PR at K step S
template <int LAYER, int KH, int S, int PR, int NT>
__device__ __forceinline__ void load_pair(const uint8_t *smem, const int (&pb)[NT], u32x4 (&x)[2][3]) {
  constexpr int plane = LAYER == 1 ? kY0Plane : kY1Plane;
  constexpr int off = step_offset<LAYER, KH, S>();
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) x[h][pl] = *reinterpret_cast<const u32x4 *>(smem + pb[2 * PR + h] + off + pl * plane);
}

}  // namespace
}  // namespace dx

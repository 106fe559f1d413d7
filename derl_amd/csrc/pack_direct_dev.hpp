// Between two updates of a native epoch (dx_cnn_ppo_epoch), on the default route -- uint8 frames, factored tail, every conv
// stage on the bf16 matrix cores -- nothing reads the fp32 weight mirrors: the forward (convstack_train.hip), the rollout
// (convstack.hip), the data gradients (dgrad_b6.hip) and the first layer's weight gradient (conv0_b16.hip) read bf16 planes
// in their own orders, the weight gradients read no weights at all.  These pieces write those planes straight from the
// canonical parameters (derl/models.py:94-124's state_dict: OIHW), where dx_cnn_pack takes four launches (mirrors, planes,
// two re-orderings): the exact three-term split of the same fp32 values, so every plane is bit-identical to dx_cnn_pack's.
// Round 5: no launch of their own any more -- they run in extra workgroups of launch_tail_pack's first kernel (tail.hip).
//   thread = one 16-byte piece (8 consecutive k of one row) of one of five images:
//   conv0 planes [3][32][256]            k = (kh, kw, c)                  (conv0_b16.hip, the conv-stack kernels)
//   Wf1 [wave][8 steps][3][64 lanes][8]   row oc, k = (kh, kw, ic)          (launch_convstack_pack's order)
//   Wf2 [wave][9 steps][3][64][8]         row oc, k = (kh, kw, ic)
//   Wd1 [wave = 2 parity + ic tile][8][3][64][8]   row ic, k = (a, b, oc), tap (py + 2 a, px + 2 b)   (dgrad_b6.hip)
//   Wd2 [wave = ic tile + 4 K half][9][3][64][8]   row ic, k = (kh, kw, oc)
#pragma once
#include "bf16_split.hpp"
#include "igemm.hpp"

namespace dx {
namespace {

constexpr int kN0 = 32 * 256 / 8;                  // pieces of a conv0 plane
constexpr int kN1 = 8 * 8 * 64, kN2 = 8 * 9 * 64;  // (wave, step, lane) triples of conv1 / conv2
constexpr int kPackDirectPieces = kN0 + 2 * (kN1 + kN2);

struct PackDirectArgs {
  const float *w0, *w1, *w2;  // canonical OIHW: (32, 4, 8, 8), (64, 32, 4, 4), (64, 64, 3, 3)
  uint16_t *p0, *f1, *f2, *d1, *d2;
};

__device__ __forceinline__ void store3(uint16_t *dst, long long plane_stride, const float (&v)[8]) {
  const Split4 a = split4(f32x4{v[0], v[1], v[2], v[3]}), b = split4(f32x4{v[4], v[5], v[6], v[7]});
  *reinterpret_cast<u32x4 *>(dst) = u32x4{a.hi.x, a.hi.y, b.hi.x, b.hi.y};
  *reinterpret_cast<u32x4 *>(dst + plane_stride) = u32x4{a.mid.x, a.mid.y, b.mid.x, b.mid.y};
  *reinterpret_cast<u32x4 *>(dst + 2 * plane_stride) = u32x4{a.lo.x, a.lo.y, b.lo.x, b.lo.y};
}

// piece q (one 16-byte piece of one of the five images); q >= kPackDirectPieces: nothing
__device__ __forceinline__ void pack_direct_piece(const PackDirectArgs &a, int q) {
  float v[8];
  if (q < kN0) {  // conv0: row oc, k = 8 q' .. : (kh, kw, c) with 4 channels -> two (kh, kw) taps of 4 channels
    const int oc = q >> 5, k0 = 8 * (q & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j, c = k & 3, tap = k >> 2;  // tap = kh * 8 + kw
      v[j] = a.w0[(oc * 4 + c) * 64 + tap];
    }
    store3(a.p0 + oc * 256 + k0, 32 * 256, v);
    return;
  }
  q -= kN0;
  const int lane = q & 63, n16 = lane & 15, kq = lane >> 4;
  if (q < kN1) {  // Wf1: wave (nt, kh2), step s: tap 8 kh2 + s, ic 8 kq ..
    const int s = (q >> 6) & 7, wave = q >> 9, nt = wave & 3, kh2 = wave >> 2;
    const int oc = 16 * nt + n16, tap = 8 * kh2 + s;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w1[(oc * 32 + 8 * kq + j) * 16 + tap];
    store3(a.f1 + ((wave * 8 + s) * 3) * 512 + lane * 8, 512, v);
    return;
  }
  q -= kN1;
  if (q < kN2) {  // Wf2: K step g = 9 kh2 + s: tap g / 2, ic 32 (g % 2) + 8 kq ..
    const int r = q >> 6, s = r % 9, wave = r / 9, nt = wave & 3, kh2 = wave >> 2;
    const int oc = 16 * nt + n16, g = 9 * kh2 + s, tap = g >> 1, ic0 = 32 * (g & 1) + 8 * kq;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w2[(oc * 64 + ic0 + j) * 9 + tap];
    store3(a.f2 + ((wave * 9 + s) * 3) * 512 + lane * 8, 512, v);
    return;
  }
  q -= kN2;
  if (q < kN1) {  // Wd1: wave = 2 p + ic tile, step s: tap (a, b) = s / 2, oc 32 (s % 2) + 8 kq ..
    const int s = (q >> 6) & 7, wave = q >> 9, p = wave >> 1, ict = wave & 1, py = p >> 1, px = p & 1;
    const int ic = 16 * ict + n16, ab = s >> 1, kh = py + 2 * (ab >> 1), kw = px + 2 * (ab & 1), oc0 = 32 * (s & 1) + 8 * kq;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w1[((oc0 + j) * 32 + ic) * 16 + kh * 4 + kw];
    store3(a.d1 + ((wave * 8 + s) * 3) * 512 + lane * 8, 512, v);
    return;
  }
  q -= kN1;
  if (q < kN2) {  // Wd2: wave = ic tile + 4 K half, step g = 9 kh2 + s: tap g / 2, oc 32 (g % 2) + 8 kq ..
    const int r = q >> 6, s = r % 9, wave = r / 9, nt = wave & 3, kh2 = wave >> 2;
    const int ic = 16 * nt + n16, g = 9 * kh2 + s, tap = g >> 1, oc0 = 32 * (g & 1) + 8 * kq;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w2[((oc0 + j) * 64 + ic) * 9 + tap];
    store3(a.d2 + ((wave * 9 + s) * 3) * 512 + lane * 8, 512, v);
  }
}

}  // namespace
}  // namespace dx

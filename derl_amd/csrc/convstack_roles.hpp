// Role-specialised conv-stack kernels (convstack_train.hip: the forward of a training minibatch; convstack.hip's
// convstack_roll_kernel: the rollout step / act): what the B waves (conv1 + conv2 over the whole contraction, streaming
// weight fragments) and the A waves (conv0 of the next image, software-pipelined) run.  derl/models.py:104-111.
#pragma once
#include "convstack_dev.hpp"

namespace dx {
namespace {

// A whole layer's contraction for this wave's 16 channels: 2 NS K steps (two halves of NS: the fragment buffer R holds
// one half) over NT pixel tiles, two tiles at a time; weight fragments R[step % NS].  A wave is ALONE on its SIMD's
// matrix pipe for most of this, so nothing but its own instruction stream hides its LDS latency: the activation
// fragments (three planes per tile, convstack_dev.hpp: load_pair) are read TWO tile pairs ahead of the MFMAs that use
// them (one pair ahead left 24 reads of the four B waves -- in lockstep behind the barrier -- landing together under
// eleven MFMAs: 13,100 cycles for 9,200 of matrix time).  The NEXT weights ride along: once global step G - 1 has issued
// its last MFMA (at the first tile pair of step G) its three registers are reloaded -- half 0's step with half 1's
// (`nexta`), half 1's with the next layer's first half (`nextb`, NNEXT steps).
// (V: timing variants of the diag flavour, WRONG results -- 1: no weight reloads in the loops, 2: no LDS re-reads of the
// activation fragments, 4: A multiplies nothing)
template <int V, int LAYER, int NT, int NS, int NNEXT, int DIST, int G, int PR>
__device__ __forceinline__ void conv_run_from(const uint8_t *smem, const int (&pb)[NT], u32x4 (&R)[9][3], f32x4 (&acc)[NT],
                                              u32x4 (&x0)[2][3], u32x4 (&x1)[2][3], const uint16_t *nexta, const uint16_t *nextb,
                                              unsigned lane_bytes) {
  constexpr int kPairs = NT / 2, kAll = 2 * NS * kPairs, kThis = G * kPairs + PR;  // position in the (step, pair) sequence
  constexpr int S = G % NS;
  constexpr bool last = kThis == kAll - 1;
  constexpr int kAhead = kThis + DIST, GA = kAhead / kPairs, PA = kAhead % kPairs;  // the pair read now (DIST = 2, or 1 where registers are short)
  u32x4 x2[2][3];
  constexpr bool reads = kAhead < kAll && !(V & 2);
  constexpr bool weights = PR == 0 && G >= 1 && !(V & 1) && (G - 1 < NS || (G - 1) % NS < NNEXT);
  // One scheduling region per tile pair: its twelve MFMAs, the six fragment reads of the pair DIST ahead and (first pair
  // of a step) the three weight loads of the step that has just retired -- INTERLEAVED, one memory instruction behind each
  // of the first MFMAs.  Issued as a burst (mac_first | reads | the other eleven MFMAs: convstack.hip's form, fine with
  // two such waves per SIMD) every read and load holds this wave's in-order stream while the LDS / the texture path
  // takes it -- 16 cycles for a 1-KB global load -- and no MFMA issues meanwhile.
  if constexpr (reads) load_pair<LAYER, GA / NS, GA % NS, PA, NT>(smem, pb, x2);
  if constexpr (weights) {
    constexpr int SP = (G - 1) % NS;  // the step that has just retired
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) R[SP][pl] = load_piece(G - 1 < NS ? nexta : nextb, (SP * 3 + pl) * 512, lane_bytes);
  }
  acc[2 * PR] = mac_first(acc[2 * PR], R[S], x0[0]);
  acc[2 * PR] = mac_rest(acc[2 * PR], R[S], x0[0]);
  acc[2 * PR + 1] = mac_rest(mac_first(acc[2 * PR + 1], R[S], x0[1]), R[S], x0[1]);
  if constexpr (reads) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA,
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
    }
  }
  if constexpr (weights) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one global load
    }
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 2 * kTerms - (reads ? 6 : 0) - (weights ? 3 : 0), 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!last) {
    constexpr int GN = PR + 1 < kPairs ? G : G + 1, PN = PR + 1 < kPairs ? PR + 1 : 0;
    if constexpr (V & 2) conv_run_from<V, LAYER, NT, NS, NNEXT, DIST, GN, PN>(smem, pb, R, acc, x1, x0, nexta, nextb, lane_bytes);
    else if constexpr (DIST == 1) conv_run_from<V, LAYER, NT, NS, NNEXT, DIST, GN, PN>(smem, pb, R, acc, x2, x2, nexta, nextb, lane_bytes);
    else conv_run_from<V, LAYER, NT, NS, NNEXT, DIST, GN, PN>(smem, pb, R, acc, x1, x2, nexta, nextb, lane_bytes);
  }
}

template <int V, int LAYER, int NT, int NS, int NNEXT, int DIST = 2>
__device__ __forceinline__ void conv_run(const uint8_t *smem, const int (&pb)[NT], u32x4 (&R)[9][3], f32x4 (&acc)[NT],
                                         const uint16_t *nexta, const uint16_t *nextb, unsigned lane_bytes) {
  static_assert(NT % 2 == 0 && NS <= 9 && NNEXT <= 9, "tiles go in pairs; the fragment buffer holds nine steps");
  constexpr int kPairs = NT / 2;
  u32x4 x0[2][3], x1[2][3];
  load_pair<LAYER, 0, 0, 0, NT>(smem, pb, x0);
  if constexpr (DIST == 2) load_pair<LAYER, 0, 1 / kPairs, 1 % kPairs, NT>(smem, pb, x1);  // position 1 of the (step, pair) sequence
  conv_run_from<V, LAYER, NT, NS, NNEXT, DIST, 0, 0>(smem, pb, R, acc, x0, x1, nexta, nextb, lane_bytes);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = NS - 1; s < NNEXT; ++s)  // the last step's registers (and conv2's ninth step behind conv1's eight)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      if constexpr (!(V & 1)) R[s][pl] = load_piece(nextb, (s * 3 + pl) * 512, lane_bytes);
}

// bias + ReLU + exact three-way split of four channels (oc0 ..) of conv1's output pixel p: the y1 planes in LDS, and
// the fp32 values kept for the backward
__device__ __forceinline__ void finish_conv1(uint8_t *smem, f32x4 sum, f32x4 bias, int p, int oc0, float *gy1) {
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x = sum[j] + bias[j];
    v[j] = relu_keep_nan(x);
  }
  if (p < kP1) {
    store_planes4(smem, oY1 + (p / 9) * kY1R + (p % 9) * kY1P + oc0 * 2, kY1Plane, v);
    if (gy1) *reinterpret_cast<f32x4 *>(gy1 + p * 64 + oc0) = v;  // (training: kept for the backward)
  }
}

}  // namespace
}  // namespace dx

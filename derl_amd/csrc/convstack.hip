// The rollout's conv stack -- and, with the policy's tail inside, the whole act step, or T steps against the synthetic
// device env -- as ONE launch with ONE workgroup per env: conv(4->32, k8, s4) + ReLU -> conv(32->64, k4, s2) + ReLU ->
// conv(64->64, k3, s1) + ReLU of derl/models.py:104-111 (the batched Policy.act forward of derl/policies.py:61-80, inside
// derl/runners/env_runner.py:41-69's loop), for 84 x 84 x 4 uint8 frames.
//
// Why: at rollout sizes (32-256 images) the layer-by-layer kernels (igemm_lat.hip: one 32x32 tile per workgroup) stream
// every operand of every tile through L2 and ran at 0.21-0.29 of the fp32-MFMA peak, three dependent launches per step.
// The whole conv stack of an image is image-local, and 256 envs are 256 CUs: a workgroup keeps ITS env's frame, y0
// (20 x 20 x 32) and y1 (9 x 9 x 64) in LDS as exact three-term bf16 planes, never writes them to memory, and the step's
// chain frame -> conv stack -> y2 Wc^T -> sample -> next frame never leaves the CU.
//
// ALL THREE layers run on the bf16 matrix cores at fp32 accuracy (convstack_dev.hpp):
//   * conv0: a uint8 pixel is exact in bf16, the fp32 weight splits EXACTLY into three bf16 terms (hi + mid + lo,
//     pre-split by dx_cnn_pack), every product is exact in fp32, v_mfma_f32_32x32x16_bf16 accumulates in fp32; the 1/255
//     is applied once to the finished sum;
//   * conv1 / conv2: BOTH operands are split exactly into three bf16 terms (activations when the previous layer's
//     epilogue writes them to LDS, weights by dx_cnn_pack) and x w = sum of nine exact products, of which the six
//     largest are multiplied (v_mfma_f32_16x16x32_bf16, smallest terms first); the three dropped ones are below 2^-23 of
//     the product (kTerms = 9 multiplies them too).
// Round 5: the eight waves are SPECIALISED (convstack_roll_kernel below; convstack_train.hip is the training forward's
// twin and explains the roles, the four barriers per step and the measurements behind them).  The one-role-for-all
// kernel of round 4 -- every wave running conv0 | epilogue | conv1 (K halves) | exchange | conv2 | exchange in lockstep:
// 43,000 cycles per step for 22,000 of matrix time -- is gone: 18.4 -> 14.4 us per step at 32 envs, 18.9 -> 17.0 at 256.
// Results equal the layer-by-layer path to fp32 rounding (other summation order).
#include "bf16_split.hpp"
#include "heads_dev.hpp"
#include "igemm.hpp"
#include "synth_dev.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "convstack_roles.hpp"

namespace dx {
namespace {

// The sample of step t from the eight waves' partial outputs (one wave; heads.hip: tail_act_block's rule):
// out[j] = the waves' sums in wave order + beff[j]; softmax over the A logits, inverse-CDF draw, log-prob, value.
template <int NW = 8>
__device__ __forceinline__ void sample_step(const ConvStackArgs &a, const float *tailred, int t, int e, int lane) {
  const int A = a.A, col = lane & 31;
  const float *part = tailred + (t & 1) * (8 * kTailOut);
  float x = 0.f;
  if (lane <= A) {  // output `lane`: the waves' sums in wave order + beff
#pragma unroll
    for (int w = 0; w < NW; ++w) x += part[w * kTailOut + lane];
    x += a.beff[lane];
  }
  float mx = -INFINITY;
  for (int k = 0; k < A; ++k) mx = fmaxf(mx, lane_value(x, k));
  const bool is_logit = col < A;
  const float ex = is_logit ? expf(x - mx) : 0.f;
  float accs = 0.f, cdf = 0.f;
  for (int k = 0; k < A; ++k) {  // sequential float32 running sum in column order
    accs += lane_value(ex, k);
    if (k == col) cdf = accs;
  }
  const float u = a.uniforms ? a.uniforms[e] : uniform01(a.seed, a.counter + t, a.env0 + e);
  const float thresh = u * accs;
  const unsigned long long below = __ballot(is_logit && cdf <= thresh);
  int act = __popcll(below & 0xffffffffull);
  if (act > A - 1) act = A - 1;
  const float lse = mx + logf(accs);
  const float la = lane_value(x, act) - lse;
  const float val = lane_value(x, A);
  if (lane == 0) {
    const long long row = static_cast<long long>(t) * a.row_stride + e;
    a.actions[row] = act;
    a.log_prob[row] = la;
    a.values[row] = val;
  }
}

// The rollout step / act with the waves SPECIALISED as in convstack_train.hip (read its header first) and the steps of
// the horizon pipelined: waves 0-3 (B) run conv1 (tiles 0-3) and conv2 of step t over the whole contraction -- no K-half
// exchange -- one conv0 tile of step t + 1, and the policy's tail (y2 Wc^T: four waves' partial sums instead of eight);
// waves 4-7 (A) run conv1's tiles 4-5, generate the synthetic env's next frame (a hash: all vector ALU, done while the
// B waves still multiply), conv0 of step t + 1 under B's conv2, and the sample of step t - 1.  Same LDS map, four
// barriers per step (alpha .. delta).  T = 1 without env: one act step (dx_cnn_act), the A waves then only help with conv1.
// The next frame does not wait for this step's action: the measurement env ignores it -- with an
// action-dependent device env the step would serialise sample -> frame -> conv0.
__global__ __launch_bounds__(512) void convstack_roll_kernel(const ConvStackArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool roleB = wave < 4;
  const int e = blockIdx.x;  // the env
  const int T = a.T;
  const bool env = a.env != 0;
  const long long step_bytes = static_cast<long long>(a.row_stride) * kFrameB;  // frames of one step of the whole batch
  unsigned long long tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int stamp_wave = kDiag ? (a.env0 >> 24) & 7 : 0;
  const int stamp_step = kDiag ? a.stamp_step : 0;
  int t = 0;
#define DX_CS_MARK(i) if (kDiag && a.stamps && t == stamp_step) tk[i] = __builtin_amdgcn_s_memtime();
  if (kDiag && a.stamps) { tk[14] = __builtin_amdgcn_s_memrealtime(); tk[15] = __builtin_amdgcn_s_memtime(); }

  // ---- step 0's frame and conv0's weight planes (resident for the whole launch): all eight waves ----
  {
    const uint8_t *src = a.obs + static_cast<long long>(e) * kFrameB;
    u32x4 fr[4];  // the frame: 1,764 pieces of 16 bytes
#pragma unroll
    for (int u = 0; u < 4; ++u) fr[u] = *reinterpret_cast<const u32x4 *>(src + 16 * min(tid + 512 * u, kFrameB / 16 - 1));
    u32x4 wv[6];  // conv0's planes: 3 x 32 rows x 32 pieces
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = u * 512 + tid;
      wv[u] = *reinterpret_cast<const u32x4 *>(a.Wb0 + (i >> 10) * 8192 + ((i >> 5) & 31) * 256 + (i & 31) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (tid + 512 * u < kFrameB / 16) put_frame_unit(smem, tid + 512 * u, fr[u]);
    bias0_to_lds(smem, a.bias0, tid);
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = u * 512 + tid;
      *reinterpret_cast<u32x4 *>(smem + oW0 + (i >> 10) * kWPlaneB + ((i >> 5) & 31) * kWRowB + (i & 31) * 16) = wv[u];
    }
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const unsigned lane16 = static_cast<unsigned>(lane * 16);
  float *tailred = reinterpret_cast<float *>(smem + oTail);  // [step parity][8 waves][kTailOut outputs] (waves 0-3 write)
  auto load_bias0 = [&](f32x4 (&bias0)[4], int l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bias0[q] = *reinterpret_cast<const f32x4 *>(smem + oBias0 + 4 * (8 * q + 4 * (l >> 5)));
  };
  // 16-byte unit `unit` of the synthetic env's frame after step t (synth_atari_block's hash of (seed, counter, position))
  auto frame_unit = [&](int t_, int unit) {
    const uint64_t key = synth_mix64(a.env_seed * 0x9E3779B97F4A7C15ull + (a.env_counter + t_));
    const uint64_t p = static_cast<uint64_t>(a.env0 + e) * (kFrameB / 16) + unit;
    const uint64_t x = synth_mix64(key + 2 * p * 0x9E3779B97F4A7C15ull);
    const uint64_t y = synth_mix64(key + (2 * p + 1) * 0x9E3779B97F4A7C15ull);
    return u32x4{static_cast<uint32_t>(x), static_cast<uint32_t>(x >> 32), static_cast<uint32_t>(y), static_cast<uint32_t>(y >> 32)};
  };
  // ... into the rollout buffer, and (when a step follows) into the LDS slot its conv0 reads
  auto put_unit = [&](int t_, int unit, u32x4 v, bool to_lds) {
    if (unit < kFrameB / 16) {
      *reinterpret_cast<u32x4 *>(a.obs + (t_ + 1) * step_bytes + static_cast<long long>(e) * kFrameB + 16 * unit) = v;
      if (to_lds) put_frame_unit(smem, unit, v);
    }
  };

  if (roleB) {
    // ============ B: conv1 (tiles 0-3), conv2, the tail of step t; conv0's tile nt of step t + 1 ============
    const int nt = wave;
    const int kq = lane >> 4, oc0 = 16 * nt + 4 * kq;
    const uint16_t *w1h0 = a.Wf1 + nt * (8 * 3 * 512), *w1h1 = a.Wf1 + (nt + 4) * (8 * 3 * 512);
    const uint16_t *w2h0 = a.Wf2 + nt * (9 * 3 * 512), *w2h1 = a.Wf2 + (nt + 4) * (9 * 3 * 512);
    u32x4 R[9][3];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) R[s][pl] = load_piece(w1h0, (s * 3 + pl) * 512, lane16);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) R[8][pl] = R[7][pl];
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(a.bias1 + oc0);
    const f32x4 bias2 = *reinterpret_cast<const f32x4 *>(a.bias2 + oc0);
    const int Jp = (a.A + 8) & ~7;
    lds_barrier();  // p1: frame 0 and conv0's planes are in LDS
    // conv0: the A waves take tiles 0 .. 11 (three each); the half-valid 13th tile (pixels 384 .. 399) is shared by waves 0
    // and 1, eight K chunks each, wave 1's sums reach wave 0 through LDS behind the phase's last barrier (convstack_train.hip)
    auto conv0_half = [&](f32x16 (&accb)[1], int l) {
      if (nt == 0) conv0_mfma<1, 4, 1, 0, 8>(smem, 12, l, accb);
      else if (nt == 1) {
        conv0_mfma<1, 4, 1, 8, 16>(smem, 12, l, accb);
        exch_put(smem, l, accb[0]);
      }
    };
    auto conv0_finish = [&](f32x16 (&accb)[1], int l) {
      if (nt == 0) {
        exch_add(smem, l, accb[0]);
        f32x4 bias0[4];
        load_bias0(bias0, l);
        conv0_store<1, 4, 1>(smem, 12, l, accb, bias0, nullptr);
      }
    };
    {
      f32x16 accb[1];
      conv0_half(accb, lane);
      lds_barrier();  // p2: every wave has read the frame
      conv0_finish(accb, lane);
    }
    for (t = 0; t < T; ++t) {
      const bool next0 = env && t + 1 < T;  // (uniform) another step follows: its conv0 runs under this step's conv2
      DX_CS_MARK(7)
      lds_barrier();  // alpha: y0 of this step is complete
      DX_CS_MARK(0)
      int pb1[4];
      {
        const int l1 = opaque(lane), m16 = l1 & 15, k4 = l1 >> 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int p = min(16 * mt + m16, kP1 - 1), oy = p / 9, ox = p - 9 * oy;
          pb1[mt] = oY0 + 2 * oy * kY0R + 2 * ox * kY0P + 16 * k4;
        }
      }
      f32x4 acc[4] = {zero4, zero4, zero4, zero4};
      conv_run<0, 1, 4, 8, 9>(smem, pb1, R, acc, w1h1, w2h0, lane16);
      u32x4 fr[4];  // pieces 12 + nt + 4 u of the env's next frame (the A waves make pieces 0 .. 11: their conv1 tiles end
                    // later than these four -- a SIMD serves its older wave first -- so the B waves take the larger share)
      if (env) {
        const int lf = opaque(lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) fr[u] = frame_unit(t, (12 + nt + 4 * u) * 64 + lf);
      }
      DX_CS_MARK(1)
      lds_barrier();  // beta: every wave has read y0 -- the y1 planes may overwrite its start, the next frame its end
      DX_CS_MARK(2)
#pragma unroll
      for (int m = 0; m < 4; ++m) finish_conv1(smem, acc[m], bias1, 16 * m + (opaque(lane) & 15), oc0, nullptr);
      if (env) {
        const int lf = opaque(lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) put_unit(t, (12 + nt + 4 * u) * 64 + lf, fr[u], next0);
      }
      DX_CS_MARK(3)
      lds_barrier();  // gamma: y1 and the next frame are complete
      DX_CS_MARK(4)
      int pb2[4];
      {
        const int l2 = opaque(lane), m16 = l2 & 15, k4 = l2 >> 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int p = min(16 * mt + m16, kP2 - 1), oy = p / 7, ox = p - 7 * oy;
          pb2[mt] = oY1 + oy * kY1R + ox * kY1P + 16 * k4;
        }
      }
      f32x4 acc2[4] = {zero4, zero4, zero4, zero4};
      conv_run<0, 2, 4, 9, 8>(smem, pb2, R, acc2, w2h1, w1h0, lane16);
      f32x16 accb[1];
      if (next0) conv0_half(accb, opaque(lane));
      DX_CS_MARK(5)
      lds_barrier();  // delta: every wave has read y1 and the frame -- the y0 planes may overwrite both
      DX_CS_MARK(6)
      const int l3 = opaque(lane), n16 = l3 & 15;
      f32x4 v2[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v2[m][j] = relu_keep_nan(acc2[m][j] + bias2[j]);
        if (a.y2 && 16 * m + n16 < kP2)
          *reinterpret_cast<f32x4 *>(a.y2 + static_cast<long long>(e) * (kP2 * 64) + (16 * m + n16) * 64 + oc0) = v2[m];
      }
      if (a.Wc) {
        // ---- the policy's tail: out[j] = sum over (pixel, channel) of y2 Wc[j] + beff[j].  Fragment order of the copy:
        // [wave' = nt + 4 (tile / 2)][tile % 2][row j of Jp][lane][4] (launch_tail_pack) ----
        const unsigned ow = static_cast<unsigned>(l3 * 16);
        float mine = 0.f;
        for (int grp = 0; 4 * grp <= a.A; ++grp) {  // groups of four outputs (uniform trip count)
          f32x4 wc[4][4];
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              wc[j][m] = 4 * grp + j <= a.A
                             ? __builtin_bit_cast(f32x4, load16(a.Wc + (((nt + 4 * (m >> 1)) * 2 + (m & 1)) * Jp + 4 * grp + j) * 256, ow))
                             : zero4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (4 * grp + j > a.A) continue;  // uniform
            float part = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              float d = v2[m][0] * wc[j][m][0];
              d = __builtin_fmaf(v2[m][1], wc[j][m][1], d);
              d = __builtin_fmaf(v2[m][2], wc[j][m][2], d);
              d = __builtin_fmaf(v2[m][3], wc[j][m][3], d);
              part += 16 * m + n16 < kP2 ? d : 0.f;
            }
            const float tot = wave_sum_all(part);
            mine = l3 == 4 * grp + j ? tot : mine;
          }
        }
        if (l3 <= a.A) tailred[(t & 1) * (8 * kTailOut) + nt * kTailOut + l3] = mine;  // sampled by wave 7 behind the next alpha
      }
      if (next0) conv0_finish(accb, l3);
    }
  } else {
    // ============ A: conv1's tiles 4-5 of step t; the env's next frame; conv0 of step t + 1; the sample of step t - 1 ============
    const int aw = wave - 4;
    const int kq = lane >> 4, oc0 = 16 * aw + 4 * kq;
    const uint16_t *w1h0 = a.Wf1 + aw * (8 * 3 * 512), *w1h1 = a.Wf1 + (aw + 4) * (8 * 3 * 512);
    u32x4 R[9][3];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) R[s][pl] = load_piece(w1h0, (s * 3 + pl) * 512, lane16);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) R[8][pl] = R[7][pl];
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(a.bias1 + oc0);
    // conv0's tiles aw, 4 + aw, 8 + aw: one instantiation per wave (the addresses fold against the constant tile index)
    auto conv0_tiles = [&](int l, f32x16 (&acc0)[3]) {
      switch (aw) {
        case 0: conv0_mfma<3, 4, 3>(smem, 0, l, acc0); break;
        case 1: conv0_mfma<3, 4, 3>(smem, 1, l, acc0); break;
        case 2: conv0_mfma<3, 4, 3>(smem, 2, l, acc0); break;
        default: conv0_mfma<3, 4, 3>(smem, 3, l, acc0); break;
      }
    };
    auto conv0_tiles_store = [&](int l, const f32x16 (&acc0)[3], const f32x4 (&bias0)[4]) {
      switch (aw) {
        case 0: conv0_store<3, 4, 3>(smem, 0, l, acc0, bias0, nullptr); break;
        case 1: conv0_store<3, 4, 3>(smem, 1, l, acc0, bias0, nullptr); break;
        case 2: conv0_store<3, 4, 3>(smem, 2, l, acc0, bias0, nullptr); break;
        default: conv0_store<3, 4, 3>(smem, 3, l, acc0, bias0, nullptr); break;
      }
    };
    lds_barrier();  // p1
    {
      f32x16 acc0[3];
      conv0_tiles(lane, acc0);
      lds_barrier();  // p2
      f32x4 bias0[4];
      load_bias0(bias0, lane);
      conv0_tiles_store(lane, acc0, bias0);
    }
    for (t = 0; t < T; ++t) {
      const bool next0 = env && t + 1 < T;
      f32x16 acc0[3];
      DX_CS_MARK(7)
      lds_barrier();  // alpha
      DX_CS_MARK(0)
      if (a.Wc && t > 0 && aw == 3) sample_step<4>(a, tailred, t - 1, e, opaque(lane));  // (its partial sums were complete before alpha)
      int pb1[2];
      {
        const int l1 = opaque(lane), m16 = l1 & 15, k4 = l1 >> 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int p = min(16 * (4 + mt) + m16, kP1 - 1), oy = p / 9, ox = p - 9 * oy;
          pb1[mt] = oY0 + 2 * oy * kY0R + 2 * ox * kY0P + 16 * k4;
        }
      }
      f32x4 acc[2] = {zero4, zero4};
      // The synthetic env's NEXT frame of this env: this wave's three KB of it, in registers until beta frees the LDS slot.
      // Generated BEFORE the conv1 tiles: this SIMD's B wave is served first by the matrix pipe, so these tiles end when B's
      // four do whenever they start, and the hash's vector-ALU work runs beside B's MFMAs instead of behind both
      u32x4 fr[3];  // pieces aw + 4 u (the B waves make pieces 12 .. 27 behind their four conv1 tiles)
      if (env) {
        const uint64_t key = synth_mix64(a.env_seed * 0x9E3779B97F4A7C15ull + (a.env_counter + t));
        const int lf = opaque(lane);
#pragma unroll
        for (int u = 0; u < 3; ++u) fr[u] = frame_unit(t, (aw + 4 * u) * 64 + lf);
        if (tid == 256) {
          const uint64_t r = synth_mix64(~key + static_cast<uint64_t>(a.env0 + e) * 0xD1B54A32D192ED03ull);
          const float u0 = static_cast<float>(r & 0xffffff) * (1.0f / 16777216.0f);
          const float u1 = static_cast<float>((r >> 24) & 0xffffff) * (1.0f / 16777216.0f);
          const long long row = static_cast<long long>(t) * a.row_stride + e;
          if (a.rewards) a.rewards[row] = u0 < a.p_reward ? ((r >> 63) ? -1.f : 1.f) : 0.f;
          if (a.resets) a.resets[row] = u1 < a.p_reset ? 1 : 0;
        }
      }
      conv_run<0, 1, 2, 8, 0, 1>(smem, pb1, R, acc, w1h1, w1h0, lane16);
      DX_CS_MARK(1)
      lds_barrier();  // beta
      DX_CS_MARK(2)
#pragma unroll
      for (int m = 0; m < 2; ++m) finish_conv1(smem, acc[m], bias1, 16 * (4 + m) + (opaque(lane) & 15), oc0, nullptr);
      if (env) {
        const int lf = opaque(lane);
#pragma unroll
        for (int u = 0; u < 3; ++u) put_unit(t, (aw + 4 * u) * 64 + lf, fr[u], next0);
      }
      DX_CS_MARK(3)
      lds_barrier();  // gamma
      DX_CS_MARK(4)
      if (next0) {
        const int l0 = opaque(lane);
        conv0_tiles(l0, acc0);
#pragma unroll
        for (int s = 0; s < 8; ++s)  // the next step's conv1 taps 0-7: under the epilogue
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) R[s][pl] = load_piece(w1h0, (s * 3 + pl) * 512, lane16);
      }
      DX_CS_MARK(5)
      lds_barrier();  // delta
      DX_CS_MARK(6)
      if (next0) {
        const int l0 = opaque(lane);
        f32x4 bias0[4];
        load_bias0(bias0, l0);
        conv0_tiles_store(l0, acc0, bias0);
      }
    }
  }
  if (a.Wc) {  // the last step's sample
    lds_barrier();
    if (wave == 7) sample_step<4>(a, tailred, T - 1, e, lane);
  }
#undef DX_CS_MARK
  if (kDiag && a.stamps && tid == 64 * stamp_wave) {
    tk[8] = __builtin_amdgcn_s_memtime();
    tk[13] = __builtin_amdgcn_s_memrealtime();
    tk[15] = tk[8] - tk[15];
    tk[14] = tk[13] - tk[14];
    for (int i = 0; i < 16; ++i) a.stamps[blockIdx.x * 16 + i] = tk[i];
  }
}

// conv1 / conv2's bf16 planes and the tail's Wc in the order the kernel's waves read them: one thread per
// 16-byte piece.  Wave (nt = wave & 3: 16 output channels, kh2 = wave >> 2: K half); lane (n16 = row of the
// tile, kq = k group): the A fragment of v_mfma_f32_16x16x32_bf16 at K step s is 8 consecutive k of row n16.
constexpr int kPieces1 = 8 * 8 * 3 * 64, kPieces2 = 8 * 9 * 3 * 64, kPiecesC = 8 * 2 * 8 * 64;
__global__ __launch_bounds__(256) void convstack_pack_kernel(const uint16_t *Wb1, const uint16_t *Wb2, const float *Wc, uint16_t *Wf1,
                                                             uint16_t *Wf2, float *Wcf) {
  int q = blockIdx.x * 256 + threadIdx.x;
  const int lane = q & 63, n16 = lane & 15, kq = lane >> 4;
  if (q < kPieces1 + kPieces2) {
    const bool second = q >= kPieces1;
    if (second) q -= kPieces1;
    const int ns = second ? 9 : 8, K = second ? 576 : 512;
    int r = q >> 6;
    const int pl = r % 3;
    r /= 3;
    const int s = r % ns, wave = r / ns, nt = wave & 3, kh2 = wave >> 2;
    const uint16_t *src = (second ? Wb2 : Wb1) + pl * (64 * K) + (16 * nt + n16) * K + (ns * kh2 + s) * 32 + 8 * kq;
    reinterpret_cast<u32x4 *>(second ? Wf2 : Wf1)[q] = *reinterpret_cast<const u32x4 *>(src);
    return;
  }
  q -= kPieces1 + kPieces2;
  if (q >= kPiecesC || Wc == nullptr) return;
  // the tail: lane holds channels oc0 .. oc0 + 3 of pixel 16 (2 kh2 + m) + n16 of y2; output row j
  const int r = q >> 6, j = r & 7, m = (r >> 3) & 1, wave = r >> 4, nt = wave & 3, kh2 = wave >> 2;
  const int p = 16 * (2 * kh2 + m) + n16, oc0 = 16 * nt + 4 * kq;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (p < kP2) v = *reinterpret_cast<const f32x4 *>(Wc + j * (kP2 * 64) + p * 64 + oc0);
  reinterpret_cast<f32x4 *>(Wcf)[q] = v;
}

}  // namespace

// (the Wc copy: room for 24 padded output rows -- up to 18 actions + the value, launch_tail_pack writes it)
long long convstack_pack_elems(int which) { return which == 0 ? kPieces1 * 8LL : which == 1 ? kPieces2 * 8LL : kPiecesC * 4LL * (kTailOut / 8); }

int launch_convstack_pack(const uint16_t *Wb1, const uint16_t *Wb2, const float *Wc, uint16_t *Wf1, uint16_t *Wf2, float *Wcf,
                          hipStream_t stream) {
  DX_REQUIRE(Wb1 && Wb2 && Wf1 && Wf2 && (Wc == nullptr || Wcf), "convstack_pack: bad arguments");
  DX_REQUIRE(aligned(Wb1, 16) && aligned(Wb2, 16) && aligned(Wf1, 16) && aligned(Wf2, 16) && aligned(Wc, 16) && aligned(Wcf, 16),
             "convstack_pack: operands must be 16-byte aligned");
  const int pieces = kPieces1 + kPieces2 + (Wc ? kPiecesC : 0);
  hipLaunchKernelGGL(convstack_pack_kernel, dim3((pieces + 255) / 256), dim3(256), 0, stream, Wb1, Wb2, Wc, Wf1, Wf2, Wcf);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

bool convstack_supported(int in_h, int in_w, int in_c) { return in_h == kIn && in_w == kIn && in_c == 4; }

// One launch of the conv-stack kernel (igemm.hpp: ConvStackArgs): y2 only, a whole act step (policy tail and
// sampling in the kernel), or T steps against the synthetic device env.
int launch_convstack(const ConvStackArgs &args, hipStream_t stream) {
  ConvStackArgs a = args;
  DX_REQUIRE(a.obs && a.Wb0 && a.bias0 && a.Wf1 && a.bias1 && a.Wf2 && a.bias2 && a.B >= 1 && a.T >= 1, "convstack: bad arguments");
  DX_REQUIRE(aligned(a.obs, 16) && aligned(a.Wb0, 16) && aligned(a.Wf1, 16) && aligned(a.Wf2, 16) && aligned(a.bias0, 16) &&
                 aligned(a.bias1, 16) && aligned(a.bias2, 16) && (a.y2 == nullptr || aligned(a.y2, 16)) &&
                 (a.Wc == nullptr || aligned(a.Wc, 16)),
             "convstack: frames, weight planes, biases, Wc and y2 must be 16-byte aligned");
  DX_REQUIRE(a.y2 != nullptr || a.Wc != nullptr, "convstack: neither y2 nor the tail requested");
  DX_REQUIRE(a.Wc == nullptr || (a.beff && a.actions && a.log_prob && a.values && a.A >= 1 && a.A + 1 <= 19),
             "convstack: the in-kernel tail needs beff, the three outputs and <= 18 actions");
  DX_REQUIRE(a.env == 0 ? a.T == 1 : (a.Wc != nullptr && a.row_stride >= a.B && a.uniforms == nullptr),
             "convstack: T > 1 only against the synthetic env, with the tail in the kernel");
  DX_REQUIRE(a.train ? (a.env == 0 && a.Wc == nullptr && a.y0 && a.y1 && a.y2 && aligned(a.y0, 16) && aligned(a.y1, 16))
                     : (a.sample_idx == nullptr && a.y0 == nullptr && a.y1 == nullptr),
             "convstack: a training forward stores y0, y1 and y2 and has no tail; a rollout step takes no gather");
  if (a.row_stride < a.B) a.row_stride = a.B;
  a.stamps = nullptr;
  a.stamp_step = 0;
  int cus = 0;
  if (int rc = device_cus(&cus)) return rc;
  // a training forward: one workgroup per CU (the LDS holds one image), each walks its share of the minibatch
  const int B = a.train ? (a.B < cus ? a.B : cus) : a.B;  // workgroups
  if (a.train) return launch_convstack_train(a, B, stream);  // waves specialised by layer, two images in flight
  {
    DX_LDS_OPT_IN(convstack_roll_kernel, kLdsBytesX);
#if DX_DIAG
    if (getenv("DX_CS_DIAG")) {  // in-kernel phase cycles of one step (DX_CS_STEP) as seen by wave DX_CS_DIAG, on stderr (synchronous)
      unsigned long long *dev_stamps = nullptr;
      DX_HIP(hipMalloc(&dev_stamps, static_cast<size_t>(B) * 128));
      a.stamps = dev_stamps;
      const int stamp_wave = atoi(getenv("DX_CS_DIAG")) & 7;
      a.env0 |= stamp_wave << 24;  // (timing build only: sampling positions are not what is looked at)
      a.stamp_step = getenv("DX_CS_STEP") ? atoi(getenv("DX_CS_STEP")) : 0;
      if (a.stamp_step >= a.T) a.stamp_step = a.T - 1;
      hipLaunchKernelGGL(convstack_roll_kernel, dim3(B), dim3(512), kLdsBytesX, stream, a);
      DX_LAUNCH_CHECK();
      DX_HIP(hipStreamSynchronize(stream));
      std::vector<unsigned long long> h(static_cast<size_t>(B) * 16);
      DX_HIP(hipMemcpy(h.data(), dev_stamps, h.size() * 8, hipMemcpyDeviceToHost));
      DX_HIP(hipFree(dev_stamps));
      static const int order[8] = {7, 0, 1, 2, 3, 4, 5, 6};
      const bool b = stamp_wave < 4;
      const char *what[7] = {"wait at alpha (y0 complete)", b ? "conv1 loop, tiles 0-3" : "sample (wave 7), conv1 tiles 4-5, next frame's hash",
                             "wait at beta (y0 read by all)", b ? "bias / ReLU / split -> y1" : "y1 of tiles 4-5, next frame -> LDS + rollout buffer",
                             "wait at gamma (y1 and frame complete)", b ? "conv2 loop + one conv0 tile" : "conv0 tiles of the next step",
                             "wait at delta (y1 and frame dead)"};
      double total = 0;
      fprintf(stderr, "[convstack_roll B=%d T=%d step %d wave %d] cycles per workgroup (mean):\n", B, a.T, a.stamp_step, stamp_wave);
      for (int i = 0; i < 7; ++i) {
        double d = 0;
        for (int k = 0; k < B; ++k) d += static_cast<double>(h[k * 16 + order[i + 1]] - h[k * 16 + order[i]]) / B;
        total += d;
        fprintf(stderr, "  %-58s %8.0f\n", what[i], d);
      }
      fprintf(stderr, "  %-58s %8.0f\n", "total (alpha wait .. delta passed; the tail / epilogue follow)", total);
      double cyc = 0, ticks = 0;
      for (int k = 0; k < B; ++k) { cyc += static_cast<double>(h[k * 16 + 15]); ticks += static_cast<double>(h[k * 16 + 14]); }
      fprintf(stderr, "  whole launch: %.0f cycles per workgroup (%.0f per step) in %.2f us: shader clock %.0f MHz\n", cyc / B, cyc / B / a.T,
              ticks / B / 100.0, cyc / ticks * 100.0);
      return DX_OK;
    }
#endif
    hipLaunchKernelGGL(convstack_roll_kernel, dim3(B), dim3(512), kLdsBytesX, stream, a);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
}

}  // namespace dx

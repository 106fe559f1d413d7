// The forward of a TRAINING minibatch through the conv stack of derl/models.py:104-111 as one launch, one workgroup
// per CU walking its share of the images -- the image-resident arithmetic of convstack_dev.hpp with the eight waves
// SPECIALISED and two images in flight (convstack.hip's convstack_roll_kernel is the rollout's twin):
//
//   waves 0-3 ("B": one per SIMD, the older wave of each SIMD pair): conv1 (pixel tiles 0-3) and conv2 of image i, and
//     ONE conv0 tile of image i + 1.  Wave b owns output channels 16 b .. 16 b + 15 of both layers over the WHOLE
//     contraction (16 taps / 18 steps), so nothing is exchanged between waves: one accumulator per tile from the first
//     tap to the bias.  Its weight fragments live in ONE buffer of 9 x 3 fragments (108 registers) that cycles through
//     conv1 taps 0-7 -> taps 8-15 -> conv2 steps 0-8 -> steps 9-17 -> the next image's conv1 taps 0-7: K step s of a half
//     is overwritten by step s of the next half right after its last MFMA has issued (three 1-KB loads from the
//     fragment-ordered copies in L2, each behind one MFMA of step s + 1; no load bursts between the layers).
//   waves 4-7 ("A"): conv1's pixel tiles 4-5 of image i for the channels of their SIMD's B wave (their own 96-register
//     fragment buffer), and conv0 of image i + 1: 2-3 of its 13 pixel tiles multiply UNDER B's conv2 loop on the same
//     matrix pipes (the frame lies in LDS as bf16, converted once when it is written: no vector-ALU work in the
//     loop), and the epilogue (1 / 255, bias, ReLU, the exact three-way split into the y0 planes: the most vector-ALU
//     work of the whole stack) runs beside B's conv2 epilogue.  The next frame travels through registers: fetched from
//     HBM (a gather by sample_idx) under the conv1 loops -- 4 KB per A wave, 3 KB per B wave -- and written to LDS when
//     beta frees the slot (fetched by LDS-DMA behind beta it took 4,700 cycles and B waited for it).
//
// Why (stamps and same-box A/B: profiles/r05_ab_convstack.txt).  Round 4's one-role-for-all kernel ran every phase
// serially in all eight waves -- conv0 MFMA | conv0 epilogue | conv1 | K-half exchange + epilogue | conv2 | exchange +
// epilogue -- 42,400 cycles per image of which 22,200 are matrix time: the pipes idle through every epilogue and
// exchange, and the K halves of a SIMD pair finish their loops 3,700 cycles apart (the older wave wins the arbitration).
// Three measurements shaped this kernel: (1) ONE wave issues v_mfma_f32_16x16x32_bf16 every 16.5 cycles, TWO waves of
// a SIMD together every 12.4 (tools/ubench/mfma_issue.hip; 32x32x16: 32 against 24) -- so both waves of a SIMD should
// multiply in every phase, which is why A takes a third of conv1 and B one conv0 tile; (2) a burst of LDS reads or
// global loads holds a wave's in-order stream while the LDS / the texture path takes it and no MFMA issues meanwhile
// (conv1 alone on a SIMD: 13,600 cycles, 9,250 without the fragment re-reads, 6,400 without re-reads and weight loads)
// -- so every read and load is pinned behind its own MFMA (sched_group_barrier, convstack_roles.hpp); (3) conv0 of a
// whole image in ONE wave per SIMD beside B's conv2 took 16,300 cycles for 4,600 of matrix time: its tiles are spread
// 3 / 2 / 2 / 2 over the A waves + one per B wave (13,700), and the byte -> bf16 conversion (12 instructions per tile
// and chunk) moved out of the loop to where the frame is written (13,000).  The phase stays issue-bound: two streams
// of MFMAs, LDS reads and loads on one SIMD -- 10,000 cycles of matrix time at the two-wave rate take 13,000-14,000.
// Result: 33,800 cycles per image, 570-575 us at minibatch 8192 where round 4's kernel takes 621 on the same box.
//
// LDS map = convstack_dev.hpp's (149 KB): conv0's weight planes resident; region B holds y0 (image i) -> y1 (image i,
// at its start) + the frame of image i + 1 (at its end) -> y0 (image i + 1).  Four workgroup barriers per image:
//   alpha  y0(i) complete                 B: conv1 tiles 0-3 of image i          A: conv1 tiles 4-5; frame(i + 1) -> registers
//   beta   every wave has read y0(i)      B: bias / ReLU / split -> y1(i)        A: the same for its tiles; frame -> LDS
//   gamma  y1(i), frame(i + 1) complete   B: conv2 of image i, conv0 tile b      A: conv0 tiles 4 + a, 8 + a (12) of image i + 1
//   delta  y1(i) and the frame are dead   B: bias / ReLU -> y2(i); its y0 tile   A: epilogue -> y0(i + 1)
// Arithmetic: the same exact splits and the same six products per fp32 product as everywhere (kTerms); a tile's sum
// runs over all taps in one accumulator, so results differ from the layer-by-layer stages in the last bits (summation
// order), like every other route.  ReLU keeps a NaN (relu_keep_nan).
#include "convstack_roles.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace dx {
namespace {

template <int V>
__global__ __launch_bounds__(512) void convstack_train_kernel(const ConvStackArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool roleB = wave < 4;
  const int grid = static_cast<int>(gridDim.x);
  int e = blockIdx.x;  // the image: blockIdx + t grid
  const int steps = (a.B - static_cast<int>(blockIdx.x) + grid - 1) / grid;
  unsigned long long tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int stamp_wave = kDiag ? (a.env0 >> 24) & 7 : 0;
  const int stamp_step = kDiag ? a.stamp_step : 0;
  int t = 0;
#define DX_CS_MARK(i) if (kDiag && a.stamps && t == stamp_step) tk[i] = __builtin_amdgcn_s_memtime();
  if (kDiag && a.stamps) { tk[14] = __builtin_amdgcn_s_memrealtime(); tk[15] = __builtin_amdgcn_s_memtime(); }

  // ---- image 0's frame and conv0's weight planes (resident for the whole launch): all eight waves ----
  {
    const uint8_t *src = a.obs + static_cast<long long>(a.sample_idx ? a.sample_idx[e] : e) * kFrameB;
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
  // the gather index of the NEXT image is read one image ahead (a scalar load whose round trip sat in front of the frame
  // loads at the top of every image's conv1 phase)
  auto frame_row = [&](int img) { return img < a.B ? (a.sample_idx ? a.sample_idx[img] : img) : 0; };
  int row_next = frame_row(e + grid);

  if (roleB) {
    // =================================== B: conv1 + conv2 of image t ===================================
    const int nt = wave;  // output channels 16 nt .. 16 nt + 15 of both layers
    const int n16 = lane & 15, kq = lane >> 4, oc0 = 16 * nt + 4 * kq;
    // fragment-ordered copies (launch_convstack_pack): piece (wave' = nt + 4 half, step, plane) is one KB
    const uint16_t *w1h0 = a.Wf1 + nt * (8 * 3 * 512), *w1h1 = a.Wf1 + (nt + 4) * (8 * 3 * 512);
    const uint16_t *w2h0 = a.Wf2 + nt * (9 * 3 * 512), *w2h1 = a.Wf2 + (nt + 4) * (9 * 3 * 512);
    u32x4 R[9][3];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) R[s][pl] = load_piece(w1h0, (s * 3 + pl) * 512, lane16);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) R[8][pl] = R[7][pl];  // (defined; first written for real behind conv1's second half)
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(a.bias1 + oc0);
    const f32x4 bias2 = *reinterpret_cast<const f32x4 *>(a.bias2 + oc0);
    // conv0: the A waves take tiles 0 .. 11 (three each); the half-valid 13th tile is the first two B waves', behind their
    // conv2 loops (round 5, first version: every B wave multiplied a whole tile behind conv2 -- 13,900 cycles where conv2
    // alone takes 11,200 and the A waves with two tiles idled for 4,400)
    auto load_bias0 = [&](f32x4 (&bias0)[4], int l) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bias0[q] = *reinterpret_cast<const f32x4 *>(smem + oBias0 + 4 * (8 * q + 4 * (l >> 5)));
    };
    lds_barrier();  // p1: frame 0 and conv0's planes are in LDS
    // conv0's 13th tile (pixels 384 .. 399) is shared by waves 0 and 1, eight K chunks each (conv0_half); wave 1's sums
    // reach wave 0 through LDS behind the phase's last barrier (conv0_finish), wave 0 stores the tile
    auto conv0_half = [&](f32x16 (&accb)[1], int l) {
      if (nt == 0) conv0_mfma<1, 4, 1, 0, 8>(smem, 12, l, accb);
      else if (nt == 1) {
        conv0_mfma<1, 4, 1, 8, 16>(smem, 12, l, accb);
        exch_put(smem, l, accb[0]);
      }
    };
    auto conv0_finish = [&](f32x16 (&accb)[1], int l, float *gy0) {
      if (nt == 0) {
        exch_add(smem, l, accb[0]);
        f32x4 bias0[4];
        load_bias0(bias0, l);
        conv0_store<1, 4, 1>(smem, 12, l, accb, bias0, gy0);
      }
    };
    {
      f32x16 accb[1];
      conv0_half(accb, lane);
      lds_barrier();  // p2: every wave has read the frame
      conv0_finish(accb, lane, a.y0 + static_cast<long long>(e) * (kP0 * 32));
    }
    for (t = 0; t < steps; ++t, e += grid) {
      DX_CS_MARK(7)
      lds_barrier();  // alpha: y0 of this image is complete
      DX_CS_MARK(0)
      // (per-lane address tables are rebuilt per image behind an opaque copy of the lane index: hoisted out of the image
      // loop, the tables and everything the compiler derives from them -- one register per plane and tap beyond the
      // 64 KB immediate range -- do not fit beside the fragment buffer and spill)
      int pb1[4];  // byte address of the lane's pixel (tile mt) and k group in plane 0 of y0
      {
        const int l1 = opaque(lane), m16 = l1 & 15, k4 = l1 >> 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {  // (conv1's tiles 4 and 5 are the A wave's of this SIMD)
          const int p = min(16 * mt + m16, kP1 - 1), oy = p / 9, ox = p - 9 * oy;
          pb1[mt] = oY0 + 2 * oy * kY0R + 2 * ox * kY0P + 16 * k4;
        }
      }
      const bool more = t + 1 < steps;  // (uniform) an image follows
      u32x4 fr[3];  // pieces 16 + nt + 4 u of the next image's frame (the A waves carry pieces 0 .. 15)
      if (more) {
        const int next = e + grid;
        const uint8_t *src = a.obs + static_cast<long long>(row_next) * kFrameB;
        row_next = frame_row(next + grid);  // (for the image after: landed long before its use)
#pragma unroll
        for (int u = 0; u < 3; ++u) fr[u] = *reinterpret_cast<const u32x4 *>(src + 16 * min((16 + nt + 4 * u) * 64 + opaque(lane), kFrameB / 16 - 1));
      }
      f32x4 acc[4] = {zero4, zero4, zero4, zero4};
      conv_run<V, 1, 4, 8, 9>(smem, pb1, R, acc, w1h1, w2h0, lane16);
      DX_CS_MARK(1)
      lds_barrier();  // beta: every B wave has read y0 -- the y1 planes may overwrite its start
      DX_CS_MARK(2)
#pragma unroll
      for (int m = 0; m < 4; ++m)  // C/D layout: column (pixel) = lane & 15, rows (channels) 4 (lane >> 4) + j
        finish_conv1(smem, acc[m], bias1, 16 * m + (opaque(lane) & 15), oc0, a.y1 + static_cast<long long>(e) * (kP1 * 64));
      if (more) {  // (the end of region B: the y1 planes lie at its start)
        const int lf = opaque(lane);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int unit = (16 + nt + 4 * u) * 64 + lf;
          if (unit < kFrameB / 16) put_frame_unit(smem, unit, fr[u]);
        }
      }
      DX_CS_MARK(3)
      lds_barrier();  // gamma: y1 and the next image's frame are complete
      DX_CS_MARK(4)
      int pb2[4];  // the same in y1
      {
        const int l2 = opaque(lane), m16 = l2 & 15, k4 = l2 >> 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int p = min(16 * mt + m16, kP2 - 1), oy = p / 7, ox = p - 7 * oy;
          pb2[mt] = oY1 + oy * kY1R + ox * kY1P + 16 * k4;
        }
      }
      f32x4 acc2[4] = {zero4, zero4, zero4, zero4};
      conv_run<V, 2, 4, 9, 8>(smem, pb2, R, acc2, w2h1, w1h0, lane16);  // (then the next image's first taps; behind the last image nobody reads them)
      f32x16 accb[1];
      if (more && !(V & 4)) conv0_half(accb, opaque(lane));
      DX_CS_MARK(5)
      lds_barrier();  // delta: every wave has read y1 and the frame -- the y0 planes may overwrite both
      DX_CS_MARK(6)
      if (more) conv0_finish(accb, opaque(lane), a.y0 + static_cast<long long>(e + grid) * (kP0 * 32));
      float *out = a.y2 + static_cast<long long>(e) * (kP2 * 64);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int p = 16 * m + n16;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = acc2[m][j] + bias2[j];
          v[j] = relu_keep_nan(x);
        }
        if (p < kP2) *reinterpret_cast<f32x4 *>(out + p * 64 + oc0) = v;
      }
    }
  } else {
    // =================================== A: conv0 of image t + 1 ===================================
    const int aw = wave - 4;  // tiles 4 + aw, 8 + aw (and 12 for aw = 0) of the 13 tiles of 32 pixels; tile b is B wave b's
    // (measured and not kept: s_setprio 1 for these, the younger waves of the SIMD pairs -- they then finish conv1 in 6,300
    // cycles and their conv0 tiles in 7,700, but the B waves' loops stretch to 12,600 and 16,300: 615 us against 572)
    // conv0's bias for this lane's 16 accumulator rows (channels 8 q + 4 (lane >> 5) + j): fetched per image (L2), not
    // kept in 16 registers across the conv1 phase
    auto load_bias0 = [&](f32x4 (&bias0)[4], int l) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bias0[q] = *reinterpret_cast<const f32x4 *>(smem + oBias0 + 4 * (8 * q + 4 * (l >> 5)));
    };
    // conv1's tiles 4 and 5 for the channels of this SIMD's B wave (nt = aw), over the whole contraction: its own
    // fragment buffer cycles taps 0-7 -> taps 8-15 -> the next image's taps 0-7
    const int kq = lane >> 4, oc0 = 16 * aw + 4 * kq;
    const uint16_t *w1h0 = a.Wf1 + aw * (8 * 3 * 512), *w1h1 = a.Wf1 + (aw + 4) * (8 * 3 * 512);
    u32x4 R[9][3];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) R[s][pl] = load_piece(w1h0, (s * 3 + pl) * 512, lane16);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) R[8][pl] = R[7][pl];  // (never used: conv1 has eight steps per half)
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(a.bias1 + oc0);
    // conv0's tiles aw, 4 + aw, 8 + aw: one instantiation per wave, so that a lane's pixel / store addresses fold against
    // the constant tile index (with the index in a register the three tiles' address arithmetic spilled 87 registers)
    auto conv0_tiles = [&](int l, f32x16 (&acc0)[3]) {
      switch (aw) {
        case 0: conv0_mfma<3, 4, 3>(smem, 0, l, acc0); break;
        case 1: conv0_mfma<3, 4, 3>(smem, 1, l, acc0); break;
        case 2: conv0_mfma<3, 4, 3>(smem, 2, l, acc0); break;
        default: conv0_mfma<3, 4, 3>(smem, 3, l, acc0); break;
      }
    };
    auto conv0_tiles_store = [&](int l, const f32x16 (&acc0)[3], const f32x4 (&bias0)[4], float *gy0) {
      switch (aw) {
        case 0: conv0_store<3, 4, 3>(smem, 0, l, acc0, bias0, gy0); break;
        case 1: conv0_store<3, 4, 3>(smem, 1, l, acc0, bias0, gy0); break;
        case 2: conv0_store<3, 4, 3>(smem, 2, l, acc0, bias0, gy0); break;
        default: conv0_store<3, 4, 3>(smem, 3, l, acc0, bias0, gy0); break;
      }
    };
    lds_barrier();  // p1: frame 0 and conv0's planes are in LDS
    {
      f32x16 acc0[3];
      conv0_tiles(lane, acc0);
      lds_barrier();  // p2: every A wave has read the frame
      float *gy0 = a.y0 + static_cast<long long>(e) * (kP0 * 32);
      f32x4 bias0[4];
      load_bias0(bias0, lane);
      conv0_tiles_store(lane, acc0, bias0, gy0);
    }
    for (t = 0; t < steps; ++t, e += grid) {
      const bool more = t + 1 < steps;  // (uniform) an image follows
      f32x16 acc0[3];  // (per image: nothing of it lives across the loop's back edge)
      DX_CS_MARK(7)
      lds_barrier();  // alpha: y0 of image t is complete
      DX_CS_MARK(0)
      u32x4 fr[4];  // the next image's frame: this wave's 4 KB of it -- pieces aw + 4 u; the B waves carry pieces 16 .. 27 (a gather
                    // from HBM: 4,700 cycles when fetched by LDS-DMA behind beta, where B then waited for it; here it
                    // travels under the conv1 loops)
      if (more) {
        const int next = e + grid;
        const uint8_t *src = a.obs + static_cast<long long>(row_next) * kFrameB;
        row_next = frame_row(next + grid);  // (for the image after: landed long before its use)
#pragma unroll
        for (int u = 0; u < 4; ++u) fr[u] = *reinterpret_cast<const u32x4 *>(src + 16 * ((aw + 4 * u) * 64 + opaque(lane)));
      }
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
      conv_run<V, 1, 2, 8, 0, 1>(smem, pb1, R, acc, w1h1, w1h0, lane16);  // (one pair ahead: the frame's registers are live)
      DX_CS_MARK(1)
      lds_barrier();  // beta: y0 of image t is dead
      DX_CS_MARK(2)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        finish_conv1(smem, acc[m], bias1, 16 * (4 + m) + (opaque(lane) & 15), oc0, a.y1 + static_cast<long long>(e) * (kP1 * 64));
      if (more) {
        // the next image's frame (fetched into registers during B's conv1: see below) goes into the LDS slot conv0 reads
        // -- the end of region B; the y1 planes B writes meanwhile lie at its start
        const int lf = opaque(lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) put_frame_unit(smem, (aw + 4 * u) * 64 + lf, fr[u]);
      }
      DX_CS_MARK(3)
      lds_barrier();  // gamma: the frame is complete
      DX_CS_MARK(4)
      if (more && !(V & 4)) {
        const int l0 = opaque(lane);
        conv0_tiles(l0, acc0);
      }
      if (more) {  // the next image's conv1 taps 0-7: under the epilogue (not carried through the conv0 loop: 96 registers)
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) R[s][pl] = load_piece(w1h0, (s * 3 + pl) * 512, lane16);
      }
      DX_CS_MARK(5)
      lds_barrier();  // delta: the frame and y1 of image t are dead
      DX_CS_MARK(6)
      if (more) {
        float *gy0 = a.y0 + static_cast<long long>(e + grid) * (kP0 * 32);
        const int l0 = opaque(lane);
        f32x4 bias0[4];
        load_bias0(bias0, l0);
        conv0_tiles_store(l0, acc0, bias0, gy0);
      }
    }
  }
#undef DX_CS_MARK
  if (kDiag && a.stamps && tid == 64 * stamp_wave) {
    tk[8] = __builtin_amdgcn_s_memtime();
    tk[13] = __builtin_amdgcn_s_memrealtime();
    tk[15] = tk[8] - tk[15];   // the whole launch in shader cycles,
    tk[14] = tk[13] - tk[14];  // and in ticks of the constant 100 MHz clock
    for (int i = 0; i < 16; ++i) a.stamps[blockIdx.x * 16 + i] = tk[i];
  }
}

}  // namespace

// the forward of a training minibatch (ConvStackArgs::train): a.B images over `blocks` workgroups (one per CU)
int launch_convstack_train(const ConvStackArgs &args, int blocks, hipStream_t stream) {
  ConvStackArgs a = args;
  DX_LDS_OPT_IN(convstack_train_kernel<0>, kLdsBytesX);
#if DX_DIAG
  DX_LDS_OPT_IN(convstack_train_kernel<1>, kLdsBytesX);
  DX_LDS_OPT_IN(convstack_train_kernel<2>, kLdsBytesX);
  DX_LDS_OPT_IN(convstack_train_kernel<3>, kLdsBytesX);
  DX_LDS_OPT_IN(convstack_train_kernel<4>, kLdsBytesX);
  DX_LDS_OPT_IN(convstack_train_kernel<7>, kLdsBytesX);
#endif
#if DX_DIAG
  if (getenv("DX_CS_DIAG")) {  // in-kernel phase cycles of one image (DX_CS_STEP) as seen by wave DX_CS_DIAG, on stderr (synchronous)
    unsigned long long *dev_stamps = nullptr;
    DX_HIP(hipMalloc(&dev_stamps, static_cast<size_t>(blocks) * 128));
    a.stamps = dev_stamps;
    const int stamp_wave = atoi(getenv("DX_CS_DIAG")) & 7;
    a.env0 |= stamp_wave << 24;
    a.stamp_step = getenv("DX_CS_STEP") ? atoi(getenv("DX_CS_STEP")) : 0;
    const int steps = a.B / blocks;
    if (a.stamp_step >= steps) a.stamp_step = steps - 1;
    const int variant = getenv("DX_CS_VARIANT") ? atoi(getenv("DX_CS_VARIANT")) : 0;  // (timing variants: WRONG results)
    switch (variant) {
      case 1: hipLaunchKernelGGL(convstack_train_kernel<1>, dim3(blocks), dim3(512), kLdsBytesX, stream, a); break;
      case 2: hipLaunchKernelGGL(convstack_train_kernel<2>, dim3(blocks), dim3(512), kLdsBytesX, stream, a); break;
      case 3: hipLaunchKernelGGL(convstack_train_kernel<3>, dim3(blocks), dim3(512), kLdsBytesX, stream, a); break;
      case 4: hipLaunchKernelGGL(convstack_train_kernel<4>, dim3(blocks), dim3(512), kLdsBytesX, stream, a); break;
      case 7: hipLaunchKernelGGL(convstack_train_kernel<7>, dim3(blocks), dim3(512), kLdsBytesX, stream, a); break;
      default: hipLaunchKernelGGL(convstack_train_kernel<0>, dim3(blocks), dim3(512), kLdsBytesX, stream, a);
    }
    DX_LAUNCH_CHECK();
    DX_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(static_cast<size_t>(blocks) * 16);
    DX_HIP(hipMemcpy(h.data(), dev_stamps, h.size() * 8, hipMemcpyDeviceToHost));
    DX_HIP(hipFree(dev_stamps));
    static const int order[9] = {7, 0, 1, 2, 3, 4, 5, 6, 8};
    const bool b = stamp_wave < 4;
    const char *what[8] = {"wait at alpha (y0 complete)", b ? "conv1 loop (16 taps)" : "-", "wait at beta (y0 read by all)",
                           b ? "bias / ReLU / split -> y1" : "next frame by LDS-DMA", "wait at gamma (y1 complete, frame landed)",
                           b ? "conv2 loop (18 steps)" : "conv0 MFMAs of the next image", "wait at delta (y1 and frame dead)",
                           b ? "bias / ReLU -> y2 (one image only: to the launch's end)" : "conv0 epilogue -> y0 planes (to the launch's end)"};
    double total = 0;
    fprintf(stderr, "[convstack_train B=%d image %d wave %d] cycles per workgroup (mean):\n", a.B, a.stamp_step, stamp_wave);
    for (int i = 0; i < 7; ++i) {
      double d = 0;
      for (int k = 0; k < blocks; ++k) d += static_cast<double>(h[k * 16 + order[i + 1]] - h[k * 16 + order[i]]) / blocks;
      total += d;
      fprintf(stderr, "  %-58s %8.0f\n", what[i], d);
    }
    fprintf(stderr, "  %-58s %8.0f\n", "total (alpha wait .. delta passed)", total);
    double cyc = 0, ticks = 0;
    for (int k = 0; k < blocks; ++k) { cyc += static_cast<double>(h[k * 16 + 15]); ticks += static_cast<double>(h[k * 16 + 14]); }
    fprintf(stderr, "  whole launch: %.0f cycles per workgroup (%.0f per image) in %.2f us: shader clock %.0f MHz\n", cyc / blocks,
            cyc / blocks / steps, ticks / blocks / 100.0, cyc / ticks * 100.0);
    return DX_OK;
  }
#endif
  a.stamps = nullptr;
  a.stamp_step = 0;
  hipLaunchKernelGGL(convstack_train_kernel<0>, dim3(blocks), dim3(512), kLdsBytesX, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx

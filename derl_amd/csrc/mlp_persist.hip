// One PERSISTENT launch per epoch of the Gaussian two-net MLP actor-critic (BASELINE config 3: PPO
// HalfCheetah-shaped, 11,085 parameters, 32 minibatch updates of 4,096 samples per epoch, 320 per
// rollout) -- derl/alg/common.py:66-78 (Trainer.step) over the minibatches of
// derl/runners/onpolicy.py:44-62 with derl/runners/trajectory_transforms.py:84-92 in front of each.
//
// Why: an update is 21.6 KFLOP per sample on a 44 KB model -- under 2 us of arithmetic -- but as
// separate launches it is ~9 dependent kernels of ~5 us plus their boundaries (~75 us per update,
// 24 ms per rollout).  Here every workgroup keeps the WHOLE model (both nets, operational layout)
// in LDS for the entire epoch and its own copy of the Adam moments in a private slice of the
// workspace (L2-resident: 134 KB per workgroup; in registers they cost 72 VGPRs and the kernel
// spilled -- and a launch that needs scratch memory makes the runtime drain the queue); per minibatch:
//   A  forward -> Gaussian PPO / A2C loss -> backward for its 32-row tile(s), both nets: every GEMM
//      (32 rows x 32/64 x 32/64) as 16x16 tiles of v_mfma_f32_16x16x4_f32 (exact fp32 products,
//      fp32 accumulation) fed from LDS -- a lane reads 4 consecutive k with one ds_read_b128 (or
//      4 strided ds_read_b32 for a transposed operand) and feeds 4 MFMAs; the weight gradients are
//      formed as TRANSPOSED products so that a lane's 4 result registers are 4 consecutive columns
//      of its row (16-byte slab stores).  (The first version ran these on the vector ALUs like
//      mlp_fused.hip: 42 us per update, bound by ds_read bandwidth and the unpacked VALU rate.)
//      Partial gradients -> its slab
//   -- grid barrier --
//   B  workgroup w sums slice w of all slabs in a fixed order (deterministic), masks the padding,
//      forms its float64 partial of |g|^2; the last workgroup also reduces the loss terms
//   -- grid barrier --
//   C  EVERY workgroup reads the reduced gradient (67 KB) and applies clip + Adam to its own copy
//      of the model: all copies stay bit-identical, nothing is broadcast.
// Two grid barriers per update instead of ~9 kernel boundaries.  Cross-workgroup data (slabs,
// reduced gradient, partials) is stored write-through (sc1) and read with sc1 loads; a workgroup
// arrives at a barrier with ONE agent-scope atomic add after all its waves have drained their
// stores, and polls the counter with sc1 loads (MI355X_MICROARCH.md, inter-workgroup visibility:
// the "one lane of each storing workgroup adds to one counter" row).  Every spin is bounded, and a
// give-up is LOUD and harmless: the workgroup that gives up writes which barrier it was (1 + 2 x
// minibatch + barrier) into the timeout word; the epoch's write-back of parameters, moments and
// gradient is skipped (the caller's buffers stay untouched), its losses are NaN, and the code goes
// into a STICKY word of the workspace that makes every later launch on that workspace exit at once
// and that the host mirrors into the caller's pinned status word -- the next dx_mlp_ppo_epoch on
// it returns DX_ETIMEOUT naming the barrier (csrc/mlp_epoch.hip).  The grid is checked against the
// occupancy query when it is planned (one workgroup per CU must fit beside nothing else of this
// process: the launch is declined while a communicator exists, i.e. in multi-process runs).
//
// Results equal dx_mlp_ppo_epoch's to float32 rounding (other tile / slab partition of the sums),
// not bit for bit; the tests compare both against each other and against the CPU oracle.
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace dx {
namespace {

constexpr int kH = 64, kHeadLd = 32, kDP = 32, kR = 32, kT = 512;
constexpr int kLd0 = kDP + 4, kLd1 = kH + 4;  // row strides: 16-byte aligned, conflict-free b128 / column reads
// operational (LDS) layout of one net: W0 [64][36], W1 [64][68], W2 [32][68] (rows = this net's outputs), biases
constexpr int oW0 = 0, oW1 = oW0 + kH * kLd0, oW2 = oW1 + kH * kLd1, oB = oW2 + kHeadLd * kLd1, kNetLds = oB + 160;
// aligned (slab / reduced gradient) layout of one net: 16-byte rows; the W2 / b2 rows are this net's
// outputs (policy: mean d, value: row 0), like the operational layout
constexpr int aW0 = 0, aB0 = aW0 + kH * kDP, aW1 = aB0 + kH, aB1 = aW1 + kH * kH, aW2 = aB1 + kH,
              aB2 = aW2 + kHeadLd * kH, kNetA = aB2 + kHeadLd;
constexpr int kLogstdA = 2 * kNetA, kTotalA = kLogstdA + 32;
constexpr int kMaxMb = 64;
constexpr unsigned kSpinLimit = 1u << 21;
constexpr int kStampSlots = 16;  // A, barrier, B, barrier, C | A's stages: inputs, h1, h2, outputs, loss, three backward stages, slab
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
// v of lane l + v of lane l ^ 32, in both lanes (inline asm: hipcc 7.2 miscompiles the builtin with a uniform operand, heads.hip)
__device__ __forceinline__ float both_halves_sum(float v) {
  float w = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(w));  // v = [lo | lo], w = [hi | hi]
  return v + w;
}

static_assert(kNetA % 4 == 0 && kTotalA % 4 == 0, "aligned layout in whole vec4");

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct PersistArgs {
  const float *params_in;
  float *params, *grads, *exp_avg, *exp_avg_sq;
  long long off_w[6], off_b[6], off_logstd;
  int D, P;
  const float *obs, *actions, *old_lp, *adv, *old_v, *vt;
  float *adv_norm;
  const double *stats;  // (minibatches, 3) {sum, sumsq, n} of the raw advantages
  long long samples;
  int mbsize, nmb, mode, normalize;
  float norm_eps, cliprange, vcoef, ecoef;
  float max_norm, beta1, beta2, eps, omb1, omb2;
  float step_size[kMaxMb], bc2_sqrt[kMaxMb];
  float *loss_out, *grad_norm_out;
  int grad_norm_stride;
  float *slabs, *gral;    // [G][kTotalA] partial gradients, [kTotalA] reduced gradient
  float *moments;         // [G][2][kTotalA] every workgroup's own exp_avg / exp_avg_sq (compact index)
  unsigned *wtab;         // [G][kTotalA / 4] compact index -> LDS index of its first element | validity bits << 20 (written at epoch start)
  double *lossp, *sumsqp;  // [G][40], [G]
  unsigned *counter, *timeout, *sticky;
  unsigned *exits;         // workgroups that have left the launch: the last one out resets the barrier words
  unsigned *status_host;   // the caller's pinned status word as the device sees it (or NULL: copied behind the launch)
  unsigned spin_limit;
  int G;
  unsigned long long *stamps;  // optional (DX_MLP_PERSIST_STAMPS=1): [G][kStampSlots] 100 MHz ticks spent in A, barrier, B, barrier, C and in A's stages
};

__device__ __forceinline__ f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// write-through (sc1) stores and sc1 loads of everything another workgroup reads in this launch
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ f32x4 ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void st_d(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_d(const double *p) {
  return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT));
}

// The timeout word: 0 = no give-up so far; 1 + 2 x minibatch + barrier = the first give-up; kCommitted = workgroup 0
// decided to write the epoch's results back (a compare-and-swap from 0 at the last barrier's far side: the ONE word
// orders that decision against a give-up at the same barrier -- whichever swap lands first stands, so "gave up" and
// "stepped" exclude each other).
constexpr unsigned kCommitted = 0x7fffffffu;
__device__ __forceinline__ bool timed_out(const PersistArgs &a) {
  const unsigned w = __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return w != 0 && w != kCommitted;
}

// Grid barrier: every wave has drained its stores, one lane arrives, polls until `target` arrivals.
// `code` = 1 + 2 x minibatch + (0 | 1): what a give-up leaves in the timeout word.
__device__ __forceinline__ void grid_barrier(const PersistArgs &a, unsigned target, int *dead, unsigned code) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!*dead) {
      unsigned spins = 0;
      while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if (spins > a.spin_limit || ((spins & 1023u) == 0 && timed_out(a))) {
          unsigned expected = 0;  // the first give-up names the barrier
          __hip_atomic_compare_exchange_strong(a.timeout, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
          // kCommitted: workgroup 0 passed the epoch's LAST barrier and took the write-back decision before this
          // give-up could register -- every workgroup had arrived, the epoch is complete and stands
          if (expected != kCommitted) *dead = 1;  // give up for good: later barriers only arrive
          break;
        }
      }
    }
  }
  __syncthreads();
}
// after a barrier: did any workgroup give up so far?  (a give-up's store is complete before that
// workgroup's NEXT arrival -- s_waitcnt vmcnt(0) opens every barrier -- so whoever passed a later
// barrier sees it; a give-up at the very last barrier is the workgroup's own `dead`, or loses the
// compare-and-swap against workgroup 0's commit and does not count)
__device__ __forceinline__ bool gave_up(const PersistArgs &a, const int *dead) { return *dead != 0 || timed_out(a); }

// aligned vec4 index -> (LDS index of its first element, validity bits of its four elements,
// canonical index of its first element); the four elements are consecutive in all three layouts
struct Where { int lds; unsigned mask; long long canon; };
__device__ __forceinline__ Where locate(const PersistArgs &a, int v) {
  Where w{0, 0u, 0};
  const int e = 4 * v;
  if (e >= kLogstdA) {
    const int d = e - kLogstdA;
    w.lds = 2 * kNetLds + d;
    w.canon = a.off_logstd + d;
    for (int q = 0; q < 4; ++q) w.mask |= (d + q < a.P ? 1u : 0u) << q;
    return w;
  }
  const int net = e >= kNetA ? 1 : 0, r = e - net * kNetA;
  const int outs = net == 0 ? a.P : 1;
  const int base = net * kNetLds;
  // (selects between constant indices: a dynamically indexed kernel-argument array goes to scratch)
  const long long ow0 = net ? a.off_w[3] : a.off_w[0], ow1 = net ? a.off_w[4] : a.off_w[1],
                  ow2 = net ? a.off_w[5] : a.off_w[2], ob0 = net ? a.off_b[3] : a.off_b[0],
                  ob1 = net ? a.off_b[4] : a.off_b[1], ob2 = net ? a.off_b[5] : a.off_b[2];
  if (r < aB0) {
    const int j = r / kDP, k = r - j * kDP;
    w.lds = base + oW0 + j * kLd0 + k;
    w.canon = ow0 + static_cast<long long>(j) * a.D + k;
    for (int q = 0; q < 4; ++q) w.mask |= (k + q < a.D ? 1u : 0u) << q;
  } else if (r < aW1) {
    w.lds = base + oB + (r - aB0); w.canon = ob0 + (r - aB0); w.mask = 15u;
  } else if (r < aB1) {
    const int i = r - aW1, j = i / kH, k = i - j * kH;
    w.lds = base + oW1 + j * kLd1 + k; w.canon = ow1 + i; w.mask = 15u;
  } else if (r < aW2) {
    w.lds = base + oB + kH + (r - aB1); w.canon = ob1 + (r - aB1); w.mask = 15u;
  } else if (r < aB2) {
    const int i = r - aW2, o = i / kH, k = i - o * kH;  // row = output of this net
    w.lds = base + oW2 + o * kLd1 + k; w.canon = ow2 + static_cast<long long>(o) * kH + k;
    w.mask = o < outs ? 15u : 0u;
  } else {
    const int o = r - aB2;
    w.lds = base + oB + 2 * kH + o; w.canon = ob2 + o;
    for (int q = 0; q < 4; ++q) w.mask |= (o + q < outs ? 1u : 0u) << q;
  }
  return w;
}

// The vec4 of the aligned layout that hold at least one real parameter, numbered 0 .. n - 1 in layout order (the
// COMPACT index: config 3 has 2,870 of 4,184 -- observation columns 17 .. 31 and output rows 6 .. 31 / 1 .. 31 are
// padding).  Phase B's slices, the reduced gradient and the moments live in this index space, so the exchange and
// Adam touch no padding; the slabs keep the aligned layout (phase A's MFMA tiles dictate it).
struct Compact { int nd4, net0, net1, total, div_mul; };
__device__ __forceinline__ Compact compact_of(int D, int P) {
  Compact c;
  c.nd4 = (D + 3) >> 2;
  const int common = kH * c.nd4 + kH / 4 + kH * kH / 4 + kH / 4;
  c.net0 = common + (kH / 4) * P + ((P + 3) >> 2);
  c.net1 = common + (kH / 4) + 1;
  c.total = c.net0 + c.net1 + ((P + 3) >> 2);
  c.div_mul = (65536 + c.nd4 - 1) / c.nd4;  // x / nd4 == (x * div_mul) >> 16 for x < 64 * 8
  return c;
}
__device__ __forceinline__ int aligned_of(const Compact &c, int P, int i) {  // compact index -> aligned vec4 index
  if (i >= c.net0 + c.net1) return kLogstdA / 4 + (i - c.net0 - c.net1);
  const int net = i >= c.net0 ? 1 : 0, outs = net ? 1 : P;
  int r = i - net * c.net0;
  const int base = net * (kNetA / 4);
  if (r < kH * c.nd4) {
    const int j = (r * c.div_mul) >> 16;
    return base + (aW0 + j * kDP) / 4 + (r - j * c.nd4);
  }
  r -= kH * c.nd4;
  if (r < kH / 4) return base + aB0 / 4 + r;
  r -= kH / 4;
  if (r < kH * kH / 4) return base + aW1 / 4 + r;
  r -= kH * kH / 4;
  if (r < kH / 4) return base + aB1 / 4 + r;
  r -= kH / 4;
  if (r < (kH / 4) * outs) return base + aW2 / 4 + r;
  r -= (kH / 4) * outs;
  return base + aB2 / 4 + r;
}

// 16 k of a 16x16 tile: lane (l % 16, l / 16) holds 4 consecutive k of its A row / B column; MFMA t
// contracts k = base + 4 * (l / 16) + t over the four lane groups
__device__ __forceinline__ void mfma16(f32x4 &acc, const f32x4 a4, const f32x4 b4) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], b4[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], b4[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], b4[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], b4[3], acc, 0, 0, 0);
}
// operand with the contraction index along a row of M[idx][k]: one ds_read_b128
__device__ __forceinline__ f32x4 along_row(const float *M, int ld, int idx, int k0) { return lds4(M + idx * ld + k0); }
// operand with the contraction index down a column of M[k][idx]: four ds_read_b32
__device__ __forceinline__ f32x4 down_col(const float *M, int ld, int idx, int k0) {
  return f32x4{M[k0 * ld + idx], M[(k0 + 1) * ld + idx], M[(k0 + 2) * ld + idx], M[(k0 + 3) * ld + idx]};
}

__global__ __launch_bounds__(kT, 2) void mlp_persist_kernel(const PersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Wl = lds;                          // [2][kNetLds] + logstd[32]
  float *logstd_s = Wl + 2 * kNetLds;
  float *xs = logstd_s + 32;                // [R][36]     observations, zero-padded columns
  float *hs = xs + kR * kLd0;               // [2][R][68]  h1
  float *gs = hs + 2 * kR * kLd1;           // [2][R][68]  h2, later dL/d(pre-tanh 1)
  float *g2 = gs + 2 * kR * kLd1;           // [2][R][68]  dL/d(pre-tanh 2)
  float *ds = g2 + 2 * kR * kLd1;           // [2][R][36]  dL/d(outputs of net 0 / net 1), zero-padded columns
  float *heads = ds + 2 * kR * kLd0;        // [R][36]     means | value
  float *act = heads + kR * kLd0;           // [R][32]
  float *rowv = act + kR * 32;              // [4][R]: old log-prob, advantage, old value, value target
  double *redd = reinterpret_cast<double *>(rowv + 4 * kR);  // [64] float64 staging
  float *coef_s = reinterpret_cast<float *>(redd + 64);      // [4]
  int *dead = reinterpret_cast<int *>(coef_s + 4);
  float *sched = reinterpret_cast<float *>(dead + 4);        // [2][kMaxMb] step size, sqrt(1 - beta2^t)
  float *sig = sched + 2 * kMaxMb;                           // [3][32] sigma, sigma^2, log sigma of this minibatch
  float *norm_s = sig + 96;                                  // [2][kMaxMb] mean, sqrt(var) + eps of every minibatch's advantages
  double *lred = reinterpret_cast<double *>(norm_s + 2 * kMaxMb);  // [8 + 31][R] per-row loss terms
  float *comb = hs;  // phase B staging (activations are dead then)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int net = t >> 8, tn = t & 255, wn = (t >> 6) & 3;
  const int G = a.G, wg = blockIdx.x;
  const int P = a.P, D = a.D;
  const int outs = net == 0 ? P : 1, col = net == 0 ? 0 : P;
  const __amdgpu_buffer_rsrc_t r_slab =
      __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, static_cast<int>(static_cast<long long>(G) * kTotalA * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_gral = __builtin_amdgcn_make_buffer_rsrc(a.gral, 0, kTotalA * 4, 0x00020000);

  // a workspace poisoned by an earlier give-up: nothing runs on it again.  (The word is only ever written at the
  // END of a launch; a workgroup that STARTS so late that the others have already given up on it and finished may
  // read their word and leave at once -- the launch has failed by then, and says so.)
  if (const unsigned poisoned = __hip_atomic_load(a.sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    if (t == 0 && a.status_host) __hip_atomic_store(a.status_host, poisoned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // ---- epoch start: the model into LDS, this thread's share of the Adam moments into registers ----
  for (int i = t; i < 2 * kNetLds + 32 + kR * kLd0 + 6 * kR * kLd1 + 3 * kR * kLd0; i += kT) Wl[i] = 0.f;
  if (t == 0) {
    *dead = 0;
#pragma unroll
    for (int i = 0; i < kMaxMb; ++i) {  // constant indices: the argument struct stays in the kernarg segment
      sched[i] = a.step_size[i];
      sched[kMaxMb + i] = a.bc2_sqrt[i];
    }
  }
  if (t < a.nmb) {  // adv_apply_kernel's expression on the precomputed float64 sums, once per epoch
    float meanf = 0.f, denom = 1.f;
    if (a.normalize) {
      const double cnt = a.stats[3 * t + 2], mean = a.stats[3 * t] / cnt;
      double var = a.stats[3 * t + 1] / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      meanf = static_cast<float>(mean);
      denom = static_cast<float>(sqrt(var)) + a.norm_eps;
    }
    norm_s[t] = meanf;
    norm_s[kMaxMb + t] = denom;
  }
  __syncthreads();
  float *mom = a.moments + static_cast<long long>(wg) * 2 * kTotalA;  // this workgroup's exp_avg | exp_avg_sq
  unsigned *wtab = a.wtab + static_cast<long long>(wg) * (kTotalA / 4);
  const Compact cx = compact_of(D, P);
#pragma unroll 1
  for (int ci = t; ci < cx.total; ci += kT) {
    const int v = aligned_of(cx, P, ci);
    const Where w = locate(a, v);
    f32x4 m4{0.f, 0.f, 0.f, 0.f}, v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if ((w.mask >> q) & 1u) {
        Wl[w.lds + q] = a.params_in[w.canon + q];
        m4[q] = a.exp_avg[w.canon + q];
        v4[q] = a.exp_avg_sq[w.canon + q];
      }
    *reinterpret_cast<f32x4 *>(mom + 4 * ci) = m4;
    *reinterpret_cast<f32x4 *>(mom + kTotalA + 4 * ci) = v4;
    wtab[ci] = static_cast<unsigned>(w.lds) | (w.mask << 20);  // phase C reads it beside the moments instead of recomputing it
  }
  __syncthreads();
  // phase B's per-thread constants (the slice of the compact index this workgroup sums, its slab groups)
  const int VB = (cx.total + G - 1) / G;   // vec4 (compact index) per workgroup
  const int NG = kT / VB;                  // slab groups working side by side
  const int chunk = (G + NG - 1) / NG;     // slabs per group
  const int b_vi = t % VB, b_sg = t / VB;
  const bool b_on = b_sg < NG && wg * VB + b_vi < cx.total;
  const unsigned b_off = b_on ? 16u * static_cast<unsigned>(aligned_of(cx, P, wg * VB + b_vi)) : 0u;
  const int b_s0 = b_sg * chunk, b_s1 = b_s0 + chunk < G ? b_s0 + chunk : G;
  const bool q_on = t < VB && wg * VB + t < cx.total;
  const int q_c = wg * VB + t, q_v = q_on ? aligned_of(cx, P, q_c) : 0;
  const unsigned q_mask = q_on ? locate(a, q_v).mask : 0u;

  const float *W0 = Wl + net * kNetLds + oW0, *W1 = Wl + net * kNetLds + oW1, *W2 = Wl + net * kNetLds + oW2;
  const float *bs = Wl + net * kNetLds + oB;
  float *h1s = hs + net * kR * kLd1, *h2s = gs + net * kR * kLd1, *g2s = g2 + net * kR * kLd1, *g1s = h2s;
  float *dsn = ds + net * kR * kLd0;  // dL/d(this net's outputs)
  const int l16 = lane & 15, kq = 4 * (lane >> 4);  // MFMA lane coordinates: row / column in the tile, k group
  unsigned arrivals = 0;
  unsigned long long tk[kStampSlots] = {}, t_prev = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0, t_sub = t_prev;
#define DX_STAMP(i)                                                         \
  if (a.stamps) {                                                           \
    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();       \
    tk[i] += now_ - t_prev;                                                 \
    t_prev = now_;                                                          \
    t_sub = now_;                                                           \
  }
#define DX_SUBSTAMP(i)                                                      \
  if (a.stamps) {                                                           \
    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();       \
    tk[i] += now_ - t_sub;                                                  \
    t_sub = now_;                                                           \
  }

  // A tile's inputs wait in registers from one tile (or one minibatch) ahead: 2 + 2 floats per thread, 4 more for the
  // first kR threads.  Loaded at the top of phase A they were 3.0 us of a 17 us phase (DX_MLP_PERSIST_STAMPS).
  static_assert(kR * kDP == 2 * kT && kR * 32 == 2 * kT, "two observation / action elements per thread");
  float pre_x[2], pre_a[2], pre_r[4];
  int pre_k = -1, pre_tile = -1;
  auto fetch = [&](int fk, int ftile) {
    pre_k = fk; pre_tile = ftile;
    const long long fstart = static_cast<long long>(fk) * a.mbsize;
    const long long fend = a.samples - fstart < a.mbsize ? a.samples : fstart + a.mbsize;
    const long long frow0 = fstart + static_cast<long long>(ftile) * kR;
    const int frows = static_cast<int>(fend - frow0 < kR ? (fend - frow0 > 0 ? fend - frow0 : 0) : kR);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = t + u * kT, r = i / kDP, c = i - r * kDP;  // (kDP == 32: the same (row, column) for both arrays)
      pre_x[u] = (r < frows && c < D) ? a.obs[(frow0 + r) * D + c] : 0.f;
      pre_a[u] = (r < frows && c < P) ? a.actions[(frow0 + r) * P + c] : 0.f;
    }
    if (t < kR) {
      const bool ok = t < frows;
      pre_r[0] = (ok && a.mode == 0) ? a.old_lp[frow0 + t] : 0.f;
      pre_r[1] = ok ? a.adv[frow0 + t] : 0.f;
      pre_r[2] = (ok && a.mode == 0) ? a.old_v[frow0 + t] : 0.f;
      pre_r[3] = ok ? a.vt[frow0 + t] : 0.f;
    }
  };
  fetch(0, wg);
  // the loss stage's sums over a tile's rows: 8 + P terms, each split over 4 / 2 / 1 lanes of wave 0 (one lane per term
  // read its 32 rows one after the other: 1.4 us of dependent LDS reads per update)
  const int lparts = 8 + P <= 16 ? 4 : (8 + P <= 32 ? 2 : 1);
  const int lterm = lane / lparts, lpart = lane % lparts;

  for (int k = 0; k < a.nmb; ++k) {
    const long long start = static_cast<long long>(k) * a.mbsize;
    const int Bk = static_cast<int>(a.samples - start < a.mbsize ? a.samples - start : a.mbsize);
    const float inv_batch = 1.0f / static_cast<float>(Bk);
    const float meanf = norm_s[k], denom = norm_s[kMaxMb + k];
    if (t < P) {  // the Gaussian's per-dimension constants, once per minibatch instead of per row
      const float sigma = expf(logstd_s[t]);
      sig[t] = sigma; sig[32 + t] = sigma * sigma; sig[64 + t] = logf(sigma);
    }
    // ================= phase A: this workgroup's row tiles =================
    // accumulators of the transposed weight-gradient products: lane (l16, kq) holds 4 consecutive columns
    f32x4 aw0[2], aw1[4], aw2[2];  // dW0[16 wn + l16][16 kt + kq ..], dW1[16 it + l16][16 wn + kq ..], dW2[16 ot + l16][16 wn + kq ..]
    float ab0 = 0.f, ab1 = 0.f, ab2 = 0.f;
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) { aw0[q] = zero4; aw2[q] = zero4; }
#pragma unroll
    for (int q = 0; q < 4; ++q) aw1[q] = zero4;
    double lacc = 0.0;  // wave 0, lanes with lpart == 0: term i < 8 is loss sum i, term 8 + d is dL/dlogstd[d]
    const int ntiles = (Bk + kR - 1) / kR;
    for (int tile = wg; tile < ntiles; tile += G) {
      const long long row0 = start + static_cast<long long>(tile) * kR;
      const int rows = static_cast<int>(start + Bk - row0 < kR ? start + Bk - row0 : kR);
      if (!(pre_k == k && pre_tile == tile)) fetch(k, tile);  // (not the tile that was fetched ahead: cannot happen with the order below)
      __syncthreads();  // the previous tile's readers are done
#pragma unroll
      for (int u = 0; u < kR * kDP / kT; ++u) {
        const int i = t + u * kT, r = i / kDP, c = i - r * kDP;
        xs[r * kLd0 + c] = pre_x[u];
        act[i] = pre_a[u];
      }
      if (t < kR) {
        const bool ok = t < rows;
        float adv = pre_r[1];
        if (a.normalize && ok) {
          adv = (adv - meanf) / denom;
          a.adv_norm[row0 + t] = adv;
        }
        rowv[kR + t] = adv;
        rowv[t] = pre_r[0];
        rowv[2 * kR + t] = pre_r[2];
        rowv[3 * kR + t] = pre_r[3];
      }
      {  // the NEXT tile's rows (this minibatch's, or the next minibatch's first) are in flight under this tile's work
        int nk = k, ntile = tile + G;
        if (ntile >= ntiles) { nk = k + 1; ntile = wg; }
        if (nk < a.nmb) fetch(nk, ntile);
      }
      __syncthreads();
      DX_SUBSTAMP(5)
      // ---- forward, both nets side by side (waves 0-3 / 4-7); wave wn owns output columns 16 wn .. 16 wn + 15 ----
      {  // h1 = tanh(x W0^T + b0): 32 x 64, K = 32
        f32x4 c0 = zero4, c1 = zero4;
#pragma unroll
        for (int k0 = 0; k0 < kDP; k0 += 16) {
          const f32x4 b4 = along_row(W0, kLd0, 16 * wn + l16, k0 + kq);
          mfma16(c0, along_row(xs, kLd0, l16, k0 + kq), b4);
          mfma16(c1, along_row(xs, kLd0, 16 + l16, k0 + kq), b4);
        }
        const float bias = bs[16 * wn + l16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h1s[(kq + r) * kLd1 + 16 * wn + l16] = tanhf(c0[r] + bias);
          h1s[(16 + kq + r) * kLd1 + 16 * wn + l16] = tanhf(c1[r] + bias);
        }
      }
      __syncthreads();
      DX_SUBSTAMP(6)
      {  // h2 = tanh(h1 W1^T + b1): 32 x 64, K = 64
        f32x4 c0 = zero4, c1 = zero4;
#pragma unroll
        for (int k0 = 0; k0 < kH; k0 += 16) {
          const f32x4 b4 = along_row(W1, kLd1, 16 * wn + l16, k0 + kq);
          mfma16(c0, along_row(h1s, kLd1, l16, k0 + kq), b4);
          mfma16(c1, along_row(h1s, kLd1, 16 + l16, k0 + kq), b4);
        }
        const float bias = bs[kH + 16 * wn + l16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h2s[(kq + r) * kLd1 + 16 * wn + l16] = tanhf(c0[r] + bias);
          h2s[(16 + kq + r) * kLd1 + 16 * wn + l16] = tanhf(c1[r] + bias);
        }
      }
      __syncthreads();
      DX_SUBSTAMP(7)
      {  // outputs = h2 W2^T + b2: 32 x 32 (padded), K = 64; wave wn: rows 16 (wn >> 1) .., columns 16 (wn & 1) ..
        const int mt = wn >> 1, nt = wn & 1;
        f32x4 c0 = zero4;
#pragma unroll
        for (int k0 = 0; k0 < kH; k0 += 16)
          mfma16(c0, along_row(h2s, kLd1, 16 * mt + l16, k0 + kq), along_row(W2, kLd1, 16 * nt + l16, k0 + kq));
        const int n = 16 * nt + l16;
        if (n < outs) {
          const float bias = bs[2 * kH + n];
#pragma unroll
          for (int r = 0; r < 4; ++r) heads[(16 * mt + kq + r) * kLd0 + col + n] = c0[r] + bias;
        }
      }
      __syncthreads();
      DX_SUBSTAMP(8)
      if (wave == 0) {  // ---- Gaussian PPO / A2C loss, one lane per row (heads.hip: normal_loss_kernel) ----
        // Both halves of the wave work on the tile's 32 rows: lanes 32 .. 63 take the odd action dimensions of row lane - 32
        // (one lane per row walked all P dimensions twice -- 3.0 us of dependent LDS reads and divisions per update, round-5
        // stamps); the two halves' partial log-probabilities / entropies meet through one v_permlane32_swap.
        const int rb = lane & (kR - 1), dhalf = lane >> 5;
        const bool row_ok = rb < rows;
        const int b = dhalf == 0 ? rb : kR;  // (b < kR: the lane that writes its row's terms; rows past the tile get zeros)
        float lp = 0.f, ent = 0.f;
        for (int d = dhalf; d < P; d += 2) {
          const float diff = act[rb * 32 + d] - heads[rb * kLd0 + d];
          lp += -(diff * diff) / (2.f * sig[32 + d]) - sig[64 + d] - kHalfLog2Pi;
          ent += 0.5f + kHalfLog2Pi + sig[64 + d];
        }
        lp = both_halves_sum(lp);
        ent = both_halves_sum(ent);
        const float v = heads[rb * kLd0 + P];
        const float adv = row_ok ? rowv[kR + rb] : 0.f;
        const float vt = row_ok ? rowv[3 * kR + rb] : 0.f;
        float pl, vl, dlp, dv;
        if (a.mode == 0) {
          const float old_lp = row_ok ? rowv[rb] : 0.f;
          const float old_v = row_ok ? rowv[2 * kR + rb] : 0.f;
          const float ratio = expf(lp - old_lp);
          const float l1 = -ratio * adv;
          pl = l1;
          bool active = true;
          if (a.cliprange >= 0.f) {
            const float lo = 1.f - a.cliprange, hi = 1.f + a.cliprange;
            const float l2 = -fminf(fmaxf(ratio, lo), hi) * adv;
            pl = fmaxf(l1, l2);
            active = (l1 > l2) || (ratio >= lo && ratio <= hi);
          }
          dlp = active ? -adv * ratio * inv_batch : 0.f;
          const float dd = v - vt;
          const float e1 = dd * dd;
          vl = e1;
          bool vactive = true;
          if (a.cliprange >= 0.f) {
            const float dvo = v - old_v;
            const float vc = old_v + fminf(fmaxf(dvo, -a.cliprange), a.cliprange);
            const float e2 = (vc - vt) * (vc - vt);
            vl = fmaxf(e1, e2);
            vactive = (e1 > e2) || (fabsf(dvo) <= a.cliprange);
          }
          dv = vactive ? a.vcoef * 2.f * dd * inv_batch : 0.f;
        } else {
          pl = -lp * adv;
          dlp = -adv * inv_batch;
          const float dd = v - vt;
          vl = dd * dd;
          dv = a.vcoef * 2.f * dd * inv_batch;
        }
        for (int d = dhalf; d < P; d += 2) {  // dL/dmean_d = dlp (a - mu) / sigma^2 ; dL/dlogstd_d = sum_b dlp ((a - mu)^2 / sigma^2 - 1) - c_H
          const float diff = act[rb * 32 + d] - heads[rb * kLd0 + d];
          const float var = sig[32 + d];
          const float g = dlp * diff / var;
          double dls = 0.0;
          if (row_ok) dls = static_cast<double>(dlp) * (diff * diff / var - 1.f) - a.ecoef * inv_batch;
          ds[rb * kLd0 + d] = row_ok ? g : 0.f;  // (each (row, dimension) has exactly one lane)
          lred[(8 + d) * kR + rb] = dls;
        }
        if (b < kR) ds[kR * kLd0 + rb * kLd0] = row_ok ? dv : 0.f;  // the value net's single output
        double s[8] = {row_ok ? pl : 0.0, row_ok ? ent : 0.0, row_ok ? vl : 0.0, row_ok ? adv : 0.0,
                       row_ok ? v : 0.0,  row_ok ? vt : 0.0,  row_ok ? (double)(v - vt) * (v - vt) : 0.0,
                       row_ok ? (double)v * v : 0.0};
        if (b < kR) {
#pragma unroll
          for (int i = 0; i < 8; ++i) lred[i * kR + rb] = s[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave: the rows' terms are in LDS
        {  // lane (term, part) sums its share of the tile's rows in row order, the parts meet pairwise: a fixed order
          double tot = 0.0;
          if (lterm < 8 + P)
            for (int r = lpart * (kR / lparts); r < (lpart + 1) * (kR / lparts); ++r) tot += lred[lterm * kR + r];
          for (int o = 1; o < lparts; o <<= 1) tot += __shfl_xor(tot, o);
          lacc += tot;
        }
      }
      __syncthreads();
      DX_SUBSTAMP(9)
      // ---- backward of both nets; contraction over the tile's 32 rows for the weight gradients ----
      {  // dW2^T: T[j][o] = sum_m h2[m][j] dout[m][o]  (wave: j tile wn, o tiles 0 / 1)
#pragma unroll
        for (int m0 = 0; m0 < kR; m0 += 16) {
          const f32x4 a4 = down_col(h2s, kLd1, 16 * wn + l16, m0 + kq);
          mfma16(aw2[0], a4, down_col(dsn, kLd0, l16, m0 + kq));
          mfma16(aw2[1], a4, down_col(dsn, kLd0, 16 + l16, m0 + kq));
        }
        if (tn < kHeadLd)
          for (int r = 0; r < kR; ++r) ab2 += dsn[r * kLd0 + tn];
        // dL/d(pre-tanh 2) = (dout W2) * (1 - h2^2): 32 x 64, K = 32 (outputs, zero-padded)
        f32x4 c0 = zero4, c1 = zero4;
#pragma unroll
        for (int k0 = 0; k0 < kHeadLd; k0 += 16) {
          const f32x4 b4 = down_col(W2, kLd1, 16 * wn + l16, k0 + kq);
          mfma16(c0, along_row(dsn, kLd0, l16, k0 + kq), b4);
          mfma16(c1, along_row(dsn, kLd0, 16 + l16, k0 + kq), b4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y0 = h2s[(kq + r) * kLd1 + 16 * wn + l16], y1 = h2s[(16 + kq + r) * kLd1 + 16 * wn + l16];
          g2s[(kq + r) * kLd1 + 16 * wn + l16] = c0[r] * (1.f - y0 * y0);
          g2s[(16 + kq + r) * kLd1 + 16 * wn + l16] = c1[r] * (1.f - y1 * y1);
        }
      }
      __syncthreads();
      DX_SUBSTAMP(10)
      {  // dW1^T: T[j][i] = sum_m h1[m][j] g2[m][i]  (wave: j tile wn, i tiles 0..3)
#pragma unroll
        for (int m0 = 0; m0 < kR; m0 += 16) {
          const f32x4 a4 = down_col(h1s, kLd1, 16 * wn + l16, m0 + kq);
#pragma unroll
          for (int it = 0; it < 4; ++it) mfma16(aw1[it], a4, down_col(g2s, kLd1, 16 * it + l16, m0 + kq));
        }
        if (tn < kH)
          for (int r = 0; r < kR; ++r) ab1 += g2s[r * kLd1 + tn];
        // dL/d(pre-tanh 1) = (g2 W1) * (1 - h1^2): 32 x 64, K = 64; written over h2 (dead: every wave passed the barrier)
        f32x4 c0 = zero4, c1 = zero4;
#pragma unroll
        for (int k0 = 0; k0 < kH; k0 += 16) {
          const f32x4 b4 = down_col(W1, kLd1, 16 * wn + l16, k0 + kq);
          mfma16(c0, along_row(g2s, kLd1, l16, k0 + kq), b4);
          mfma16(c1, along_row(g2s, kLd1, 16 + l16, k0 + kq), b4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y0 = h1s[(kq + r) * kLd1 + 16 * wn + l16], y1 = h1s[(16 + kq + r) * kLd1 + 16 * wn + l16];
          g1s[(kq + r) * kLd1 + 16 * wn + l16] = c0[r] * (1.f - y0 * y0);
          g1s[(16 + kq + r) * kLd1 + 16 * wn + l16] = c1[r] * (1.f - y1 * y1);
        }
      }
      __syncthreads();
      DX_SUBSTAMP(11)
      {  // dW0^T: T[k][j] = sum_m x[m][k] g1[m][j]  (wave: j tile wn, k tiles 0 / 1)
#pragma unroll
        for (int m0 = 0; m0 < kR; m0 += 16) {
          const f32x4 b4 = down_col(g1s, kLd1, 16 * wn + l16, m0 + kq);
          mfma16(aw0[0], down_col(xs, kLd0, l16, m0 + kq), b4);
          mfma16(aw0[1], down_col(xs, kLd0, 16 + l16, m0 + kq), b4);
        }
        if (tn < kH)
          for (int r = 0; r < kR; ++r) ab0 += g1s[r * kLd1 + tn];
      }
    }
    DX_SUBSTAMP(12)
    {  // ---- this workgroup's partial gradient -> its slab (write-through, 16 bytes per store) ----
      const unsigned sb = (static_cast<unsigned>(wg) * kTotalA + net * kNetA) * 4u;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)  // T[k = 16 kt + kq + r][j = 16 wn + l16] = dW0[j][k]
        st16(r_slab, sb + (aW0 + (16 * wn + l16) * kDP + 16 * kt + kq) * 4u, aw0[kt]);
#pragma unroll
      for (int it = 0; it < 4; ++it)  // T[j = 16 wn + kq + r][i = 16 it + l16] = dW1[i][j]
        st16(r_slab, sb + (aW1 + (16 * it + l16) * kH + 16 * wn + kq) * 4u, aw1[it]);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot)  // T[j = 16 wn + kq + r][o = 16 ot + l16] = dW2[o][j]; only this net's real outputs
        if (16 * ot + l16 < outs) st16(r_slab, sb + (aW2 + (16 * ot + l16) * kH + 16 * wn + kq) * 4u, aw2[ot]);
      if (tn < kH) {
        st4(r_slab, sb + (aB0 + tn) * 4u, ab0);
        st4(r_slab, sb + (aB1 + tn) * 4u, ab1);
      }
      if (tn < kHeadLd) st4(r_slab, sb + (aB2 + tn) * 4u, ab2);
      if (wave == 0) {
        double *lp_ = a.lossp + static_cast<long long>(wg) * 40;
        if (lpart == 0 && lterm < 8 + P) st_d(lp_ + lterm, lacc);
      }
    }
    DX_SUBSTAMP(13)
    DX_STAMP(0)
    grid_barrier(a, (++arrivals) * static_cast<unsigned>(G), dead, 1u + 2u * k);
    DX_STAMP(1)

    // ================= phase B: slice `wg` of the gradient, summed over all slabs =================
    {
      f32x4 acc{0.f, 0.f, 0.f, 0.f};
      if (b_on) {
#pragma unroll 1
        for (int s = b_s0; s < b_s1; s += 6) {  // six loads in flight (config 3: a group's whole share), added in slab order
          f32x4 x[6];
#pragma unroll
          for (int u = 0; u < 6; ++u)
            x[u] = s + u < b_s1 ? ld16(r_slab, static_cast<unsigned>(s + u) * (kTotalA * 4u) + b_off) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < 6; ++u) acc += x[u];
        }
        *reinterpret_cast<f32x4 *>(comb + (b_sg * VB + b_vi) * 4) = acc;
      }
      __syncthreads();
      double sq = 0.0;
      if (q_on) {
        f32x4 g{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < NG; ++s) g += *reinterpret_cast<const f32x4 *>(comb + (s * VB + t) * 4);
        if (q_v >= kLogstdA / 4) g = f32x4{0.f, 0.f, 0.f, 0.f};  // dL/dlogstd comes from the loss partials below
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!((q_mask >> q) & 1u)) g[q] = 0.f;  // padding columns
          sq += static_cast<double>(g[q]) * g[q];
        }
        if (q_v < kLogstdA / 4) st16(r_gral, 16u * q_c, g);
      }
      if (wg == G - 1) {  // the last workgroup (shortest slice): loss terms and dL/dlogstd of the minibatch
        __syncthreads();
        double part = 0.0;
        const int j = t % 40, pg = t / 40;  // 12 groups of slabs side by side
        if (pg < 12) {
          const int ch = (G + 11) / 12, s0 = pg * ch, s1 = s0 + ch < G ? s0 + ch : G;
          for (int s = s0; s < s1; s += 4) {  // four loads in flight, added in slab order
            double x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = s + u < s1 ? ld_d(a.lossp + static_cast<long long>(s + u) * 40 + j) : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) part += x[u];
          }
        }
        double *stage = reinterpret_cast<double *>(comb);
        if (pg < 12) stage[pg * 40 + j] = part;
        __syncthreads();
        if (t < 40) {
          double tot = 0.0;
          for (int s = 0; s < 12; ++s) tot += stage[s * 40 + t];
          redd[t] = tot;
          if (t >= 8 && t - 8 < P) {
            const float dl = static_cast<float>(tot);
            st4(r_gral, (4 * (cx.net0 + cx.net1) + (t - 8)) * 4u, dl);
            sq += static_cast<double>(dl) * dl;
          }
        }
        __syncthreads();
        if (t == 0) {  // heads.hip: loss_reduce40_kernel
          const double count = Bk;
          const float policy = static_cast<float>(redd[0] / count), ent = static_cast<float>(redd[1] / count);
          const float value = static_cast<float>(redd[2] / count);
          float *out = a.loss_out + 8LL * k;
          out[0] = (policy - a.ecoef * ent) + a.vcoef * value;
          out[1] = policy; out[2] = ent; out[3] = value;
          out[4] = static_cast<float>(redd[3] / count);
          out[5] = static_cast<float>(redd[4] / count);
          out[6] = static_cast<float>(redd[5] / count);
          const double mean_v = redd[4] / count;
          const double var_v = count > 1 ? (redd[7] - count * mean_v * mean_v) / (count - 1) : 0.0;
          out[7] = static_cast<float>(1.0 - (redd[6] / count) / var_v);
        }
      }
      // this workgroup's float64 partial of |g|^2 (waves in order)
      sq = wave_sum_d(sq);
      __syncthreads();
      if (lane == 0) redd[40 + wave] = sq;
      __syncthreads();
      if (t == 0) {
        double tot = 0.0;
        for (int w = 0; w < kT / 64; ++w) tot += redd[40 + w];
        st_d(a.sumsqp + wg, tot);
      }
    }
    DX_STAMP(2)
    grid_barrier(a, (++arrivals) * static_cast<unsigned>(G), dead, 2u + 2u * k);
    DX_STAMP(3)

    // ================= phase C: clip + Adam on this workgroup's own copy of the model =================
    {
      if (wave == 0) {  // every workgroup forms the same sum in the same order (elementwise.hip: clip_coef)
        double s = 0.0;
        for (int i = lane; i < G; i += 64) s += ld_d(a.sumsqp + i);
        s = wave_sum_d(s);
        if (lane == 0) {
          const float norm = static_cast<float>(sqrt(s));
          if (wg == 0 && a.grad_norm_out) a.grad_norm_out[static_cast<long long>(a.grad_norm_stride) * k] = norm;
          float c = 1.f;
          if (a.max_norm > 0.f) {
            c = a.max_norm / (norm + 1e-6f);
            c = c < 1.f ? c : 1.f;
          }
          coef_s[0] = c;
          // the epoch's results go back to the caller only if no barrier of the epoch gave up: decided by ONE
          // compare-and-swap of the timeout word (0 -> kCommitted), so that a give-up at this last barrier either
          // registered before it (no write-back, DX_ETIMEOUT) or fails against it (the epoch stands)
          bool commit = false;
          if (k == a.nmb - 1 && wg == 0 && *dead == 0) {
            unsigned expected = 0;
            commit = __hip_atomic_compare_exchange_strong(a.timeout, &expected, kCommitted, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
          }
          coef_s[1] = commit ? 1.f : 0.f;
        }
      }
      __syncthreads();
      const float coef = coef_s[0], step_size = sched[k], inv_bc2 = 1.0f / sched[kMaxMb + k];
      const bool write_back = coef_s[1] != 0.f;
      constexpr int kBatch = 3;  // vec4 per lane per round: 9 loads in flight
#pragma unroll 1
      for (int v0 = t; v0 < cx.total; v0 += kBatch * kT) {  // compact indices
        f32x4 g4[kBatch], m4[kBatch], v4[kBatch];
        unsigned wp[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int v = v0 + u * kT < cx.total ? v0 + u * kT : t;  // (a valid address; the result is not used)
          wp[u] = wtab[v];
          g4[u] = ld16(r_gral, 16u * v);
          m4[u] = *reinterpret_cast<const f32x4 *>(mom + 4 * v);
          v4[u] = *reinterpret_cast<const f32x4 *>(mom + kTotalA + 4 * v);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int v = v0 + u * kT;
          if (v < cx.total) {
            const int w_lds = static_cast<int>(wp[u] & 0xfffffu);
            const unsigned w_mask = wp[u] >> 20;
            long long w_canon = 0;
            if (write_back) w_canon = locate(a, aligned_of(cx, P, v)).canon;  // (workgroup 0, the epoch's last update)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if ((w_mask >> q) & 1u) {  // elementwise.hip: adam_one
                const float g = g4[u][q] * coef;
                m4[u][q] = m4[u][q] * a.beta1 + g * a.omb1;
                v4[u][q] = v4[u][q] * a.beta2 + (g * g) * a.omb2;
                // hardware sqrt / reciprocal (1 ulp): the update is <= lr in size, so their error is ~1e-7 of
                // 3e-4 -- far below the float32 spacing of the parameter it is added to -- and every workgroup
                // computes the same bits; the IEEE sequences cost 5 us per update here (33 elements per lane)
                const float den = __builtin_amdgcn_sqrtf(v4[u][q]) * inv_bc2 + a.eps;
                const float pnew = Wl[w_lds + q] - step_size * (m4[u][q] * __builtin_amdgcn_rcpf(den));
                Wl[w_lds + q] = pnew;
                if (write_back) {  // epoch end: one workgroup writes the model, the moments and the clipped gradient back
                  a.params[w_canon + q] = pnew;
                  a.exp_avg[w_canon + q] = m4[u][q];
                  a.exp_avg_sq[w_canon + q] = v4[u][q];
                  a.grads[w_canon + q] = g;
                }
              }
            *reinterpret_cast<f32x4 *>(mom + 4 * v) = m4[u];
            *reinterpret_cast<f32x4 *>(mom + kTotalA + 4 * v) = v4[u];
          }
        }
      }
      __syncthreads();  // the next minibatch's forward reads the updated model
    }
    DX_STAMP(4)
  }
#undef DX_STAMP
#undef DX_SUBSTAMP
  if (a.stamps && t == 0)
    for (int i = 0; i < kStampSlots; ++i) a.stamps[wg * kStampSlots + i] = tk[i];
  if (gave_up(a, dead)) {  // a barrier gave up: NaN losses, and the workspace is poisoned for good
    if (wg == G - 1 && t < 8 * a.nmb) a.loss_out[t] = __builtin_nanf("");
    if (t == 0) {  // (every workgroup that saw the give-up writes the same word: to the workspace and to the host's status word)
      const unsigned word = __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | 0x80000000u;
      __hip_atomic_store(a.sticky, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.status_host) __hip_atomic_store(a.status_host, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // The last workgroup out leaves the workspace ready for the next epoch's launch -- barrier counter and timeout word
  // zero (the sticky word stays); together with the status word written straight to the host's pinned memory on the
  // failure paths that is one memset launch and one 4-byte copy launch per epoch less (14 us of a 1.3 ms epoch, counting
  // their boundaries).  Every workgroup has made its own gave_up() check above before it counts itself out; after a
  // give-up a workgroup may never get here, the words stay as they are, and the sticky word keeps every later launch out.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0 && __hip_atomic_fetch_add(a.exits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(G - 1)) {
    __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.timeout, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.exits, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

size_t persist_lds_bytes() {
  return sizeof(float) * (2 * kNetLds + 32) + sizeof(float) * (kR * kLd0 + 6 * kR * kLd1 + 3 * kR * kLd0 + kR * 32 + 4 * kR) + 64 * sizeof(double) + 32 + 4 * kMaxMb * sizeof(float) + 96 * sizeof(float) + 39 * kR * sizeof(double);
}

}  // namespace

// workspace: [64 B barrier words][gral kTotalA floats][slabs G x kTotalA floats][moments G x 2 x kTotalA floats]
// [lossp G x 40 doubles][sumsqp G doubles][wtab G x kTotalA / 4 words]
long long mlp_persist_workspace_bytes(int G) {
  return 64 + 4LL * kTotalA + 4LL * G * kTotalA + 8LL * G * kTotalA + 8LL * G * 40 + 8LL * G + 64 + 1LL * G * kTotalA;
}

// number of workgroups for minibatches of `mbsize` rows; 0 = not covered (the caller keeps the
// launch-per-stage epoch): needs the Gaussian head, obs_pad 32, >= 16 row tiles per minibatch
// The grid barrier needs every workgroup resident: the grid is at most one workgroup per CU AND at
// most what the occupancy query grants this kernel with its LDS, and the launch is declined while a
// communicator exists (a multi-process run: another rank's kernels may hold CUs of a shared GPU).
// Kernels of OTHER streams of this process delay a workgroup's start but finish by themselves, so
// they cannot starve the barrier; what is left is bounded by the spin limit and fails loudly.
bool comm_active();  // comm.hip

static int configure_persist_kernel() {
  DX_LDS_OPT_IN(mlp_persist_kernel, static_cast<int>(persist_lds_bytes()));
  return DX_OK;
}

int mlp_persist_workgroups(const dx_mlp_ctx *c, int mbsize, long long samples) {
  if (!c->has_logstd || c->obs_pad != kDP) return 0;
  const int tiles = (mbsize + kR - 1) / kR;
  if (tiles < 16 || (samples + mbsize - 1) / mbsize > kMaxMb) return 0;
  if (comm_active()) return 0;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  if (configure_persist_kernel() != DX_OK) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(mlp_persist_kernel), kT,
                                                   persist_lds_bytes()) != hipSuccess || per_cu < 1)
    return 0;
  int G = tiles < cus ? tiles : cus;  // one workgroup per CU at most: all of them must be resident
  if (G > 256) G = 256;
  return G;
}

static int g_last_route = -1;
int mlp_last_route() { return g_last_route; }
void mlp_note_route(int route) { g_last_route = route; }

// 0 while the caller's status word is clean; DX_ETIMEOUT with the barrier's name otherwise
int mlp_persist_check_status(const dx_mlp_epoch *e) {
  if (e->status_host == nullptr) return DX_OK;
  const unsigned code = *static_cast<const volatile unsigned *>(e->status_host) & 0x7fffffffu;
  if (code == 0) return DX_OK;
  return fail(DX_ETIMEOUT, "dx_mlp_ppo_epoch: an earlier persistent epoch on this workspace gave up at grid barrier %u of "
              "minibatch %u (not every workgroup arrived within the spin limit); that epoch's parameters, moments and "
              "gradient were NOT written back, its losses are NaN, and the workspace refuses further epochs",
              (code - 1) % 2, (code - 1) / 2);
}

int launch_mlp_persist_epoch(const dx_mlp_ctx *c, const dx_mlp_epoch *e, int G, hipStream_t stream) {
  DX_REQUIRE(e->workspace != nullptr && e->workspace_bytes >= mlp_persist_workspace_bytes(G) && aligned(e->workspace, 16),
             "dx_mlp_ppo_epoch: persistent epoch needs a 16-byte aligned workspace of %lld bytes (got %lld)",
             mlp_persist_workspace_bytes(G), e->workspace_bytes);
  DX_REQUIRE(e->global_batch <= 0, "dx_mlp_ppo_epoch: the persistent epoch is single-process (global_batch must be 0)");
  PersistArgs a;
  std::memset(&a, 0, sizeof(a));
  a.params_in = c->params; a.params = c->params; a.grads = c->grads;
  a.exp_avg = e->exp_avg; a.exp_avg_sq = e->exp_avg_sq;
  for (int i = 0; i < 6; ++i) { a.off_w[i] = c->off_w[i]; a.off_b[i] = c->off_b[i]; }
  a.off_logstd = c->off_logstd;
  a.D = c->obs_dim; a.P = c->policy_out;
  a.obs = e->obs; a.actions = static_cast<const float *>(e->actions);
  a.old_lp = e->old_log_prob; a.adv = e->advantages; a.old_v = e->old_values; a.vt = e->value_targets;
  a.adv_norm = e->adv_normalized;
  a.samples = e->samples; a.mbsize = e->mbsize;
  a.nmb = static_cast<int>((e->samples + e->mbsize - 1) / e->mbsize);
  a.mode = e->mode; a.normalize = e->normalize; a.norm_eps = e->norm_eps;
  a.cliprange = e->cliprange; a.vcoef = e->value_loss_coef; a.ecoef = e->entropy_coef;
  a.max_norm = static_cast<float>(e->max_grad_norm);
  a.beta1 = static_cast<float>(e->beta1); a.beta2 = static_cast<float>(e->beta2); a.eps = static_cast<float>(e->adam_eps);
  a.omb1 = static_cast<float>(1.0 - e->beta1); a.omb2 = static_cast<float>(1.0 - e->beta2);
  for (int k = 0; k < a.nmb; ++k) {  // dx_clip_adam_step_f32's scalars, formed in double
    const double step = static_cast<double>(e->first_step + k);
    a.step_size[k] = static_cast<float>(e->lr / (1.0 - pow(e->beta1, step)));
    a.bc2_sqrt[k] = static_cast<float>(sqrt(1.0 - pow(e->beta2, step)));
  }
  a.loss_out = e->loss_out; a.grad_norm_out = e->grad_norm_out; a.grad_norm_stride = e->grad_norm_stride;
  char *ws = static_cast<char *>(e->workspace);
  a.counter = reinterpret_cast<unsigned *>(ws);
  a.timeout = a.counter + 4;
  a.sticky = a.counter + 8;  // survives the reset at the end of every launch
  a.exits = a.counter + 12;
  a.status_host = nullptr;
  if (e->status_host) {  // a pinned word is mapped into the device's address space: the kernel stores to it directly
    void *mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, e->status_host, 0) == hipSuccess) a.status_host = static_cast<unsigned *>(mapped);
    else (void)hipGetLastError();
  }
  a.spin_limit = kSpinLimit;
#if DX_DIAG
  // (diag flavour: DX_MLP_PERSIST_SPIN_LIMIT=0 makes every workgroup that does not arrive last give
  // up at once -- how tests/test_native_epoch_gpu.py forces the failure path)
  if (const char *v = getenv("DX_MLP_PERSIST_SPIN_LIMIT")) a.spin_limit = static_cast<unsigned>(strtoul(v, nullptr, 10));
#endif
  a.gral = reinterpret_cast<float *>(ws + 64);
  a.slabs = a.gral + kTotalA;
  a.moments = a.slabs + static_cast<long long>(G) * kTotalA;
  a.lossp = reinterpret_cast<double *>(a.moments + 2LL * G * kTotalA);
  a.sumsqp = a.lossp + static_cast<long long>(G) * 40;
  a.wtab = reinterpret_cast<unsigned *>(a.sumsqp + G + 8);  // (64 bytes behind the partials)
  a.G = G;
  const bool want_stamps = DX_ENV("DX_MLP_PERSIST_STAMPS", 0) != 0;
  unsigned long long *stamps_dev = nullptr;
  if (want_stamps) {
    DX_HIP(hipMalloc(&stamps_dev, sizeof(unsigned long long) * kStampSlots * G));
    a.stamps = stamps_dev;
  }
  if (e->normalize) {  // every minibatch's {sum, sumsq, n} in one launch (bit-identical to the per-minibatch kernel)
    DX_REQUIRE(e->stats_all != nullptr, "dx_mlp_ppo_epoch: persistent epoch needs stats_all (minibatches x 3 doubles)");
    if (int rc = dx_adv_stats_segments_f32(e->advantages, nullptr, e->samples, e->mbsize, e->stats_all, stream)) return rc;
    a.stats = e->stats_all;
  }
  // (no memset: the workspace starts zeroed -- include/derl_amd.h -- and the last workgroup of every launch zeroes the
  // barrier counter and the timeout word again)
  const size_t lds = persist_lds_bytes();
  if (int rc = configure_persist_kernel()) return rc;
  hipLaunchKernelGGL(mlp_persist_kernel, dim3(G), dim3(kT), lds, stream, a);
  DX_LAUNCH_CHECK();
  mlp_note_route(1);
  if (e->status_host && a.status_host == nullptr)  // (not mapped: the sticky word follows the launch by a copy)
    DX_HIP(hipMemcpyAsync(e->status_host, a.sticky, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
  if (want_stamps) {  // measurement aid (synchronous): where an epoch's time goes, per update
    DX_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(static_cast<size_t>(kStampSlots) * G);
    DX_HIP(hipMemcpy(h.data(), stamps_dev, sizeof(unsigned long long) * kStampSlots * G, hipMemcpyDeviceToHost));
    DX_HIP(hipFree(stamps_dev));
    double mean[kStampSlots] = {}, hi[kStampSlots] = {};
    for (int w = 0; w < G; ++w)
      for (int i = 0; i < kStampSlots; ++i) {
        const double us = h[static_cast<size_t>(kStampSlots) * w + i] * 0.01 / a.nmb;
        mean[i] += us / G;
        if (us > hi[i]) hi[i] = us;
      }
    const unsigned long long *last = &h[static_cast<size_t>(kStampSlots) * (G - 1)];
    fprintf(stderr, "[mlp_persist G=%d nmb=%d] per update, mean over workgroups (max): A %.2f (%.2f) us, barrier %.2f (%.2f), B %.2f (%.2f), "
            "barrier %.2f (%.2f), C %.2f (%.2f); last workgroup: %.2f %.2f %.2f %.2f %.2f\n", G, a.nmb, mean[0], hi[0], mean[1], hi[1],
            mean[2], hi[2], mean[3], hi[3], mean[4], hi[4], last[0] * 0.01 / a.nmb, last[1] * 0.01 / a.nmb, last[2] * 0.01 / a.nmb,
            last[3] * 0.01 / a.nmb, last[4] * 0.01 / a.nmb);
    fprintf(stderr, "[mlp_persist] A's stages, mean (max): inputs %.2f (%.2f), h1 %.2f (%.2f), h2 %.2f (%.2f), outputs %.2f (%.2f), loss %.2f (%.2f), "
            "backward 2 %.2f (%.2f), backward 1 %.2f (%.2f), backward 0 %.2f (%.2f), slab %.2f (%.2f)\n", mean[5], hi[5], mean[6], hi[6],
            mean[7], hi[7], mean[8], hi[8], mean[9], hi[9], mean[10], hi[10], mean[11], hi[11], mean[12], hi[12], mean[13], hi[13]);
  }
  return DX_OK;
}

}  // namespace dx

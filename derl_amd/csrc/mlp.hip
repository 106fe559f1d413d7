// Two-net tanh MLP actor-critic (MuJoCo Gaussian policy / vector-observation categorical
// policy): C-ABI entry points, buffer layout, and the layer-by-layer path on the implicit-GEMM
// kernels (packing, forward, backward).  Observations up to 64 wide take the fused kernels of
// mlp_fused.hip instead (one launch per direction).
//
// Restates derl/models.py:224-237 (MLP: Linear-Tanh-Linear-Tanh-Linear, hidden 64-64) and
// :240-271 (MuJoCoModel: one independent MLP per output -- policy mean / logits and value --
// plus a free `logstd` parameter) and the autograd backward of derl/alg/common.py:70.
// Parameters are flat in the reference's state_dict order:
//   [logstd (P)]  module_list.0.{0,2,4}.{weight,bias}  module_list.1.{0,2,4}.{weight,bias}
// The head output is (B, 32): columns 0..P-1 policy outputs, column P the value.
#include "igemm.hpp"
#include "mlp_fused.hpp"
#include <cstdlib>
#include <cstring>

using namespace dx;

namespace {

constexpr int kH = 64, kHeadLd = 32;

// dst[r][c] = (col_off <= c < col_off + C) ? src[r*rs + (c - col_off)*cs] : 0
__global__ __launch_bounds__(256) void strided_pad_kernel(float *dst, long long R, int Cp, const float *src,
                                                          int C, long long rs, long long cs, int col_off) {
  const long long total = R * Cp;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / Cp;
    const int c = static_cast<int>(i - r * Cp) - col_off;
    dst[i] = (c >= 0 && c < C) ? src[r * rs + c * cs] : 0.f;
  }
}

int strided_pad(float *dst, long long R, int Cp, const float *src, int C, long long rs, long long cs,
                int col_off, hipStream_t s) {
  long long blocks = (R * Cp + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(strided_pad_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, dst, R, Cp,
                     src, C, rs, cs, col_off);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// slices of >= 64 rows: the slabs are tiny (<= 16 KB per slice and layer) and the wgrads are
// latency chains of 32-row steps on a handful of workgroups, so short slices win
int msplit_bound(long long M) {  // >= the msplit of every batch <= M
  long long ms = (M + 63) / 64;
  if (ms > 256) ms = 256;
  return static_cast<int>(ms < 1 ? 1 : ms);
}

// DX_MLP_UNFUSED=1: the layer-by-layer implicit-GEMM path also for narrow observations
bool use_fused(const dx_mlp_ctx *c) {
  const bool off = DX_ENV("DX_MLP_UNFUSED", 0) != 0;
  return !off && mlp_fused_supported(c->obs_pad);
}

constexpr int kFusedMaxSlabs = 128;

// slabs per layer the buffers are sized for: the split of the GEMM path or one per workgroup of
// the fused backward (8-row tiles at the smallest batches)
int slab_capacity(const dx_mlp_ctx *c) {
  int cap = msplit_bound(c->max_batch);
  if (mlp_fused_supported(c->obs_pad)) {
    const long long tiles = (static_cast<long long>(c->max_batch) + 7) / 8;
    const int fused = static_cast<int>(tiles < kFusedMaxSlabs ? tiles : kFusedMaxSlabs);
    if (fused > cap) cap = fused;
  }
  return cap;
}

MlpFusedArgs fused_args(const dx_mlp_ctx *c, int B) {
  MlpFusedArgs a;
  std::memset(&a, 0, sizeof(a));
  a.params = c->params;
  for (int i = 0; i < 6; ++i) { a.off_w[i] = c->off_w[i]; a.off_b[i] = c->off_b[i]; }
  a.D = c->obs_dim; a.Dp = c->obs_pad; a.P = c->policy_out; a.B = B;
  a.xpad = c->xpad; a.head = c->head; a.dhead = c->dhead;
  for (int n = 0; n < 2; ++n) { a.h1[n] = c->h1[n]; a.h2[n] = c->h2[n]; }
  a.slabs = c->slabs; a.slab_per_net = c->slab_per_net; a.ms_cap = slab_capacity(c);
  return a;
}

void msplit_for(long long M, int *msplit, int *mper) {
  const long long ms = msplit_bound(M);
  *mper = static_cast<int>(((M + ms - 1) / ms + 31) / 32 * 32);
  *msplit = static_cast<int>((M + *mper - 1) / *mper);
}

Gather rows_of(const void *src, int width) {
  Gather g;
  std::memset(&g, 0, sizeof(g));
  g.src = src; g.img_stride = width; g.H = g.W = 1; g.C = width;
  g.OHW = 1; g.OW = 1; g.div_img = make_fastdiv(1); g.div_row = make_fastdiv(1);
  g.sy = g.sx = 1; g.nseg = 1; g.seglen = width;
  return g;
}

NTArgs nt(const Gather &g, const float *Wp, const float *bias, float *out, int ldc, long long M, int N, int K) {
  NTArgs a;
  std::memset(&a, 0, sizeof(a));
  a.g = g; a.Wp = Wp; a.bias = bias; a.out = out; a.ldc = ldc;
  a.M = static_cast<int>(M); a.N = N; a.K = K; a.ksplit = 1;
  return a;
}

}  // namespace

extern "C" {

int dx_mlp_init(dx_mlp_ctx *c) {
  DX_TRACE("dx_mlp_init");
  DX_REQUIRE(c != nullptr, "dx_mlp_init: null ctx");
  DX_REQUIRE(c->struct_bytes == static_cast<int>(sizeof(dx_mlp_ctx)),
             "dx_mlp_init: struct size mismatch (caller %d, library %d)", c->struct_bytes,
             static_cast<int>(sizeof(dx_mlp_ctx)));
  DX_REQUIRE(c->obs_dim >= 1 && c->obs_dim <= 4096, "dx_mlp_init: obs_dim %d out of range", c->obs_dim);
  DX_REQUIRE(c->policy_out >= 1 && c->policy_out <= 31, "dx_mlp_init: need 1 <= policy outputs <= 31");
  DX_REQUIRE(c->max_batch >= 1, "dx_mlp_init: max_batch < 1");
  const int D = c->obs_dim, P = c->policy_out;
  c->obs_pad = (D + 31) / 32 * 32;
  long long off = 0;
  c->off_logstd = -1;
  if (c->has_logstd) { c->off_logstd = 0; off = P; }
  for (int net = 0; net < 2; ++net) {
    const int outs = net == 0 ? P : 1;
    const long long wsz[3] = {static_cast<long long>(kH) * D, static_cast<long long>(kH) * kH,
                              static_cast<long long>(outs) * kH};
    const long long bsz[3] = {kH, kH, outs};
    for (int l = 0; l < 3; ++l) {
      c->off_w[3 * net + l] = off; off += wsz[l];
      c->off_b[3 * net + l] = off; off += bsz[l];
    }
  }
  c->param_count = off;
  long long po = 0;
  auto take = [&](long long n) { long long o = po; po += (n + 63) / 64 * 64; return o; };
  for (int net = 0; net < 2; ++net) {
    c->pk_f0[net] = take(static_cast<long long>(kH) * c->obs_pad);
    c->pk_d2[net] = take(kH * kHeadLd);
    c->pk_d1[net] = take(kH * kH);
  }
  c->packed_count = po;
  const int ms = slab_capacity(c);
  // per net: L0 [64][obs_pad]+[64], L1 [64][64]+[64], L2 [32][64]+[32]
  c->slab_per_net = static_cast<long long>(ms) * (kH * c->obs_pad + kH + kH * kH + kH + kHeadLd * kH + kHeadLd);
  c->slab_count = 2 * c->slab_per_net;
  c->x_count = static_cast<long long>(c->max_batch) * c->obs_pad;
  c->h_count = static_cast<long long>(c->max_batch) * kH;
  c->head_count = static_cast<long long>(c->max_batch) * kHeadLd;
  return DX_OK;
}

static int check_mlp(const dx_mlp_ctx *c, const char *who, long long B, bool bwd) {
  DX_REQUIRE(c && c->struct_bytes == static_cast<int>(sizeof(dx_mlp_ctx)) && c->param_count > 0,
             "%s: ctx not initialised by dx_mlp_init", who);
  DX_REQUIRE(B >= 1 && B <= c->max_batch, "%s: batch %lld outside [1, max_batch=%d]", who, B, c->max_batch);
  DX_REQUIRE(c->params && c->packed && c->xpad && c->h1[0] && c->h1[1] && c->h2[0] && c->h2[1] && c->head,
             "%s: forward buffers not set", who);
  if (bwd)
    DX_REQUIRE(c->grads && c->dhead && c->da && c->db && c->slabs, "%s: backward buffers not set", who);
  return DX_OK;
}

// canonical parameters -> padded / transposed mirrors (after every parameter change)
int dx_mlp_pack(const dx_mlp_ctx *c, void *stream) {
  DX_TRACE("dx_mlp_pack");
  if (int rc = check_mlp(c, "dx_mlp_pack", 1, false)) return rc;
  if (use_fused(c)) return DX_OK;  // the fused kernels read the canonical parameters
  hipStream_t s = as_stream(stream);
  const int D = c->obs_dim, P = c->policy_out;
  for (int net = 0; net < 2; ++net) {
    const int outs = net == 0 ? P : 1, col = net == 0 ? 0 : P;
    // layer 0: [64][D] -> [64][obs_pad]
    if (int rc = strided_pad(c->packed + c->pk_f0[net], kH, c->obs_pad, c->params + c->off_w[3 * net], D, D, 1, 0, s))
      return rc;
    // dgrad through layer 2: dst [j][k] = W3[k - col][j] on this net's head columns, else 0
    if (int rc = strided_pad(c->packed + c->pk_d2[net], kH, kHeadLd, c->params + c->off_w[3 * net + 2], outs, 1, kH, col, s))
      return rc;
    // dgrad through layer 1: dst [j][i] = W2[i][j]
    if (int rc = strided_pad(c->packed + c->pk_d1[net], kH, kH, c->params + c->off_w[3 * net + 1], kH, 1, kH, 0, s))
      return rc;
  }
  return DX_OK;
}

// obs (B, obs_dim) float32 -> ctx->head (B, 32); keeps xpad, h1, h2 of both nets for backward
int dx_mlp_forward(const dx_mlp_ctx *c, const float *obs, int B, void *stream) {
  DX_TRACE("dx_mlp_forward");
  if (int rc = check_mlp(c, "dx_mlp_forward", B, false)) return rc;
  DX_REQUIRE(obs != nullptr, "dx_mlp_forward: null observations");
  hipStream_t s = as_stream(stream);
  const int D = c->obs_dim, P = c->policy_out;
  if (use_fused(c)) {
    MlpFusedArgs f = fused_args(c, B);
    f.obs = obs;
    return launch_mlp_forward_fused(f, s);
  }
  if (int rc = strided_pad(c->xpad, B, c->obs_pad, obs, D, D, 1, 0, s)) return rc;
  for (int net = 0; net < 2; ++net) {
    const float *w = c->params;
    NTArgs a = nt(rows_of(c->xpad, c->obs_pad), c->packed + c->pk_f0[net], w + c->off_b[3 * net], c->h1[net], kH,
                  B, kH, c->obs_pad);
    if (int rc = launch_nt(a, false, EPI_BIAS_TANH, ST_MLP_HIDDEN, s)) return rc;
    a = nt(rows_of(c->h1[net], kH), w + c->off_w[3 * net + 1], w + c->off_b[3 * net + 1], c->h2[net], kH, B, kH, kH);
    if (int rc = launch_nt(a, false, EPI_BIAS_TANH, ST_MLP_HIDDEN, s)) return rc;
    const int outs = net == 0 ? P : 1, col = net == 0 ? 0 : P;
    a = nt(rows_of(c->h2[net], kH), w + c->off_w[3 * net + 2], w + c->off_b[3 * net + 2], c->head + col, kHeadLd,
           B, outs, kH);
    if (int rc = launch_nt(a, false, EPI_BIAS, ST_MLP_OUT, s)) return rc;
  }
  return DX_OK;
}

// T steps of the Gaussian policy against the MuJoCo-shaped synthetic device env from ONE launch (mlp_fused.hip:
// mlp_rollout_synth_kernel) -- the inner loop of derl/runners/env_runner.py:43-65 for the measurement env; buffers
// bit-identical to dx_mlp_forward + dx_normal_act_f32 + dx_synth_mujoco_step per step.  DX_ENOSUP without logstd (a
// categorical MLP) or when the fused kernels do not apply (observations wider than 64, DX_MLP_UNFUSED=1).
int dx_mlp_rollout_synth(const dx_mlp_ctx *c, float *obs, int T, int N, float *actions, float *log_prob, float *values,
                         float *rewards, uint8_t *resets, uint64_t policy_seed, uint64_t policy_counter, uint64_t env_seed,
                         uint64_t env_counter, float p_reset, void *stream) {
  DX_TRACE("dx_mlp_rollout_synth");
  if (int rc = check_mlp(c, "dx_mlp_rollout_synth", N, false)) return rc;
  if (!c->has_logstd || c->off_logstd < 0 || !use_fused(c)) return fail(DX_ENOSUP, "dx_mlp_rollout_synth: Gaussian policy on the fused kernels only");
  MlpRolloutArgs a;
  std::memset(&a, 0, sizeof(a));
  a.f = fused_args(c, N);
  a.off_logstd = c->off_logstd;
  a.obs = obs; a.actions = actions; a.log_prob = log_prob; a.values = values; a.rewards = rewards; a.resets = resets;
  a.T = T; a.policy_seed = policy_seed; a.policy_counter = policy_counter; a.env_seed = env_seed; a.env_counter = env_counter;
  a.p_reset = p_reset;
  return launch_mlp_rollout_synth(a, as_stream(stream));
}

// ctx->dhead (B, 32) -> ctx->grads (every tensor except logstd, which the Gaussian loss
// kernel writes itself), using the activations of the matching dx_mlp_forward
int dx_mlp_backward(const dx_mlp_ctx *c, int B, void *stream) {
  DX_TRACE("dx_mlp_backward");
  if (int rc = check_mlp(c, "dx_mlp_backward", B, true)) return rc;
  hipStream_t s = as_stream(stream);
  const int D = c->obs_dim, P = c->policy_out, Dp = c->obs_pad;
  int ms, mper;
  msplit_for(B, &ms, &mper);
  const int ms_cap = slab_capacity(c);
  const bool fused = use_fused(c);
  if (fused) {
    MlpFusedArgs f = fused_args(c, B);
    const int tiles = cdiv(B, mlp_fused_tile_rows(B, c->obs_pad));
    ms = tiles < kFusedMaxSlabs ? tiles : kFusedMaxSlabs;
    if (ms > ms_cap) ms = ms_cap;
    f.nslab = ms;
    if (int rc = launch_mlp_backward_fused(f, s)) return rc;
  }
  PermuteJob jobs[kMaxJobs];
  int nj = 0;
  for (int net = 0; net < 2; ++net) {
    float *base = c->slabs + net * c->slab_per_net;
    float *s0w = base, *s0b = s0w + static_cast<long long>(ms_cap) * kH * Dp;
    float *s1w = s0b + static_cast<long long>(ms_cap) * kH, *s1b = s1w + static_cast<long long>(ms_cap) * kH * kH;
    float *s2w = s1b + static_cast<long long>(ms_cap) * kH, *s2b = s2w + static_cast<long long>(ms_cap) * kHeadLd * kH;
    auto tn = [&](const Gather &g, const float *G, int ldg, int N, int K, float *slab, float *bslab, int stage) {
      TNArgs t;
      std::memset(&t, 0, sizeof(t));
      t.g = g; t.G = G; t.ldg = ldg; t.slab = slab; t.bias_slab = bslab;
      t.M = B; t.N = N; t.K = K; t.msplit = ms; t.mper = mper;
      return launch_tn(t, false, stage, s);
    };
    if (!fused) {  // layer by layer on the implicit-GEMM kernels
      // layer 2 (64 -> head columns of this net)
      if (int rc = tn(rows_of(c->h2[net], kH), c->dhead, kHeadLd, kHeadLd, kH, s2w, s2b, ST_MLP_WGRAD_OUT)) return rc;
      NTArgs a = nt(rows_of(c->dhead, kHeadLd), c->packed + c->pk_d2[net], nullptr, c->da, kH, B, kH, kHeadLd);
      a.mask_src = c->h2[net];
      if (int rc = launch_nt(a, false, EPI_DTANH, ST_MLP_DGRAD, s)) return rc;
      // layer 1
      if (int rc = tn(rows_of(c->h1[net], kH), c->da, kH, kH, kH, s1w, s1b, ST_MLP_WGRAD_HID)) return rc;
      a = nt(rows_of(c->da, kH), c->packed + c->pk_d1[net], nullptr, c->db, kH, B, kH, kH);
      a.mask_src = c->h1[net];
      if (int rc = launch_nt(a, false, EPI_DTANH, ST_MLP_DGRAD, s)) return rc;
      // layer 0 (input is data: wgrad only)
      if (int rc = tn(rows_of(c->xpad, Dp), c->db, kH, kH, Dp, s0w, s0b, ST_MLP_WGRAD_HID)) return rc;
    }
    // slabs -> canonical gradients
    const int outs = net == 0 ? P : 1, col = net == 0 ? 0 : P;
    float *g = c->grads;
    auto add = [&](const float *src, long long dst_off, long long total, int D1, long long s0, long long off,
                   long long stride) {
      jobs[nj++] = PermuteJob{src, g + dst_off, total, D1, 1, 1, s0, 1, 0, 0, off, ms, stride, 0};
    };
    add(s0w, c->off_w[3 * net], static_cast<long long>(kH) * D, D, Dp, 0, static_cast<long long>(kH) * Dp);
    add(s0b, c->off_b[3 * net], kH, 1, 1, 0, kH);
    add(s1w, c->off_w[3 * net + 1], kH * kH, kH, kH, 0, kH * kH);
    add(s1b, c->off_b[3 * net + 1], kH, 1, 1, 0, kH);
    add(s2w, c->off_w[3 * net + 2], static_cast<long long>(outs) * kH, kH, kH, static_cast<long long>(col) * kH,
        kHeadLd * kH);
    add(s2b, c->off_b[3 * net + 2], outs, 1, 1, col, kHeadLd);
  }
  return launch_permute_reduce(jobs, nj, s);
}

}  // extern "C"

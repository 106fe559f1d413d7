// Nature-DQN actor-critic network on the implicit-GEMM kernels: packing, forward, backward.
//
// Restates derl/models.py:94-124 (NatureCNNBase) + :166-214 (NatureCNNModel with
// output_units=[A, 1]) and the autograd backward that derl/alg/common.py:70 triggers.
// Parameters and gradients live in flat buffers in the reference's state_dict order and
// layout (conv OIHW, linear [out][in]); the GEMMs read packed mirrors whose K axis follows
// the NHWC activations.  NOTE (models.py:112,119-124): the reference flattens NCHW, so the
// 3136-wide linear layer is permuted while packing; there is no ReLU after it.
#include "igemm.hpp"
#include "tail_greduce_dev.hpp"
#include <cstdlib>
#include <cstring>
#include <mutex>

using namespace dx;

namespace {

constexpr int kHeadLd = 32, kHid = 512, kC0 = 32, kC1 = 64, kC2 = 64;

struct Derived {
  int h0, w0, h1, w1, h2, w2, flat;
};

int conv_out(int n, int k, int s) { return (n - k) / s + 1; }

bool use_b3() {  // experiment build + DX_SPLIT_BF16=1: big NT stages on the bf16 matrix cores
#ifdef DX_EXPERIMENT_B3
  const int v = DX_ENV("DX_SPLIT_BF16", 0);
  return v != 0;
#else
  return false;
#endif
}

// DX_C0LAT_MAX_TILES: largest 32-pixel tile count of the rollout conv0 kernel.  Crossover against the
// 256-pixel-tile kernel (eight waves per tile, pre-split planes: round 3), per PPO iteration on one
// box: 32 envs 10.96 vs 11.03 ms, 64 envs 17.69 vs 17.32 ms -- the 32x32-tile kernel up to 500 tiles
// (40 images).  (Round 2, four waves per tile: B=64 9.4 vs 11.1 us, B=128 14.7 vs 11.7 us.)
int conv0_lat_max_tiles() {
  const int v = DX_ENV("DX_C0LAT_MAX_TILES", 500);
  return v;
}

bool conv0_f32() {  // DX_CONV0_F32=1: first layer on the fp32 matrix instructions (conv0.hip)
  const int v = DX_ENV("DX_CONV0_F32", 0);
  return v != 0;
}

// DX_WGRAD_DIRECT=0: conv1 / conv2 / linear-layer weight gradients on the implicit-GEMM kernel
// instead of the dedicated ones (wgrad_direct.hip, wgrad_fc.hip); DX_WGRAD_DIRECT_MIN_B: smallest
// batch routed to them
bool wgrad_direct_on() {
  const int v = DX_ENV("DX_WGRAD_DIRECT", 1);
  return v != 0;
}
// smallest batch that takes the image-resident weight-gradient kernels: the fp32 ones (wgrad_direct.hip, wgrad_fc.hip)
// want enough images for their 256-512 persistent workgroups; the bf16 ones (wgrad_b6.hip) are one workgroup per
// image below that and measured 2x the generic kernel at 128-384 images (10.7 / 12.2 us against 21.2 / 21.6 at 128)
int wgrad_direct_min_batch() {
  const int v = DX_ENV("DX_WGRAD_DIRECT_MIN_B", 512);
  return v;
}
static int conv_wgrad_min_batch() {
  const bool set = DX_ENV_SET("DX_WGRAD_DIRECT_MIN_B");
  return set || !wgrad_b6_on() ? wgrad_direct_min_batch() : 16;
}

// DX_ROLLOUT_LANES=1: the native rollout on the caller's stream only (default 2: see
// dx_cnn_rollout_synth); DX_ROLLOUT_LANE_MIN: fewest envs a lane may have (default 64)
int rollout_lanes() {
  const int v = DX_ENV("DX_ROLLOUT_LANES", 2);
  return v;
}
int rollout_lane_min() {
  const int v = DX_ENV("DX_ROLLOUT_LANE_MIN", 64);
  return v < 1 ? 1 : v;
}

// The one piece of state the library owns: a side stream and two events per device for the
// two-lane rollout, created on first use and kept for the life of the process.
constexpr int kMaxLanes = 4;
struct SideStream { hipStream_t stream[kMaxLanes - 1]; hipEvent_t fork, join[kMaxLanes - 1]; };
SideStream *side_stream() {
  static SideStream table[16];
  static bool made[16];
  static std::mutex lock;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { fail(DX_EHIP, "side_stream: no current device"); return nullptr; }
  std::lock_guard<std::mutex> guard(lock);
  if (!made[dev]) {
    SideStream &t = table[dev];
    bool ok = hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < kMaxLanes - 1; ++i)
      ok = hipStreamCreateWithFlags(&t.stream[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&t.join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      fail(DX_EHIP, "side_stream: cannot create the rollout's side streams");
      return nullptr;
    }
    made[dev] = true;
  }
  return &table[dev];
}

bool ntp_fc_fwd() {  // DX_NTP_FC_FWD=0: the linear layer's forward on nt_dma.hip's 128x128 tiles
  const int v = DX_ENV("DX_NTP_FC_FWD", 1);
  return v != 0;
}

// DX_NT_DMA=0: linear-layer forward / dgrad on the implicit-GEMM kernel instead of nt_dma.hip
bool nt_dma_on() {
  const int v = DX_ENV("DX_NT_DMA", 1);
  return v != 0;
}

const char *g_route[ST_COUNT] = {};  // kernel family of the last launch of every stage (run_stage)

// DX_FC_FACTORED=0: the linear layer and the heads as separate GEMM stages (the layer-by-layer
// association) also where the factored tail of tail.hip applies (84 x 84 frames, <= 18 actions)
bool fc_factored_env() {
  const int v = DX_ENV("DX_FC_FACTORED", 1);
  return v != 0;
}

constexpr int kBiasChunks = 256;  // row chunks of the linear layer's bias-gradient launch (512 workgroups)

int roundup(long long v, int m) { return static_cast<int>((v + m - 1) / m * m); }

// reduction split of a wgrad over its M rows: ~5 workgroups per CU (2 per CU measured 41 %
// MFMA utilisation: both resident waves of a SIMD wait at the same time), >= min_rows rows each
// (256 for the layers whose slabs are big; 64 for the heads, whose few workgroups otherwise walk
// 256 rows in 8 dependent steps: 22 us for 5 MFLOP)
long long msplit_bound(long long M, int blocks_kn, int min_rows) {
  long long ms = 1280 / blocks_kn;
  const long long cap = (M + min_rows - 1) / min_rows;
  if (ms > cap) ms = cap;
  return ms < 1 ? 1 : ms;
}

void pick_msplit(long long M, int blocks_kn, int min_rows, int *msplit, int *mper) {
  const long long ms = msplit_bound(M, blocks_kn, min_rows);
  *mper = roundup((M + ms - 1) / ms, 32);
  *msplit = static_cast<int>((M + *mper - 1) / *mper);
}

struct SlabPlan {
  int msplit, mper;
  int bsplit;  // bias-gradient partials (= msplit unless the bias comes from its own launch)
  int direct;  // conv1 / conv2: image-resident wgrad with `msplit` persistent workgroups;
               // linear layer: wgrad_fc.hip with `msplit` row slices
  long long w_off, b_off;  // offsets (floats) of the weight / bias slabs inside ctx->slabs
};

enum Layer { L_C0 = 0, L_C1, L_C2, L_FC, L_HD, L_COUNT };

struct Plan {
  SlabPlan s[L_COUNT];
  long long total;
  long long fc_cap_floats;  // floats of the linear layer's weight-slab region (the factored tail's G / Gc / s live there)
};

}  // namespace

extern "C" {

static Derived derive(const dx_cnn_ctx *c) {
  Derived d;
  d.h0 = conv_out(c->in_h, 8, 4); d.w0 = conv_out(c->in_w, 8, 4);
  d.h1 = conv_out(d.h0, 4, 2); d.w1 = conv_out(d.w0, 4, 2);
  d.h2 = conv_out(d.h1, 3, 1); d.w2 = conv_out(d.w1, 3, 1);
  d.flat = d.h2 * d.w2 * kC2;
  return d;
}

static void layer_nk(const dx_cnn_ctx *c, int layer, int *N, int *K, int *blocks_kn) {
  const int n[L_COUNT] = {kC0, kC1, kC2, kHid, kHeadLd};
  const int k[L_COUNT] = {64 * c->in_c, 16 * kC0, 9 * kC1, c->flat, kHid};
  *N = n[layer];
  *K = k[layer];
  *blocks_kn = n[layer] <= 32 ? cdiv(k[layer], 256) * cdiv(n[layer], 32)
                              : cdiv(k[layer], 128) * cdiv(n[layer], 64);
}

static long long layer_rows(const dx_cnn_ctx *c, int layer, long long B) {
  switch (layer) {
    case L_C0: return B * c->h0 * c->w0;
    case L_C1: return B * c->h1 * c->w1;
    case L_C2: return B * c->h2 * c->w2;
    default: return B;
  }
}

static bool layer_direct_supported(const dx_cnn_ctx *c, int layer) {
  if (!wgrad_direct_on()) return false;
  if (layer == L_C1) return wgrad_direct_supported(c->h0, c->w0, kC0, c->h1, c->w1, kC1, 4, 4, 2);
  if (layer == L_C2) return wgrad_direct_supported(c->h1, c->w1, kC1, c->h2, c->w2, kC2, 3, 3, 1);
  return false;
}

// Where the factored tail keeps its partial G / s, Gc and s: the head of the linear layer's slab region.  The launchers
// trust this pointer for tail_slab_floats(B) floats, so the plan of THIS batch is checked against the reserved region
// here (nullptr with the error text set otherwise).
static float *tail_scratch(const dx_cnn_ctx *c, const Plan &plan, int B) {
  const long long need = tail_slab_floats(B, c->num_actions);
  if (need > plan.fc_cap_floats) {
    fail(DX_EINVAL, "factored tail: batch %d with %d actions needs %lld floats of slab scratch, %lld are reserved "
                    "(max_batch %d)", B, c->num_actions, need, plan.fc_cap_floats, c->max_batch);
    return nullptr;
  }
  return c->slabs + plan.s[L_FC].w_off;
}

// the first layer's weight gradient on conv0_wgrad_ks.hip (84 x 84 x 4 frames; uint8 observations take it, float ones
// keep the general GEMM with the same slab count)
static bool conv0_ks_usable(const dx_cnn_ctx *c) {
  return conv0_wgrad_ks_on() && !conv0_f32() && conv0_wgrad_ks_supported(c->in_h, c->in_w, c->in_c, c->h0, c->w0);
}

static Plan make_plan(const dx_cnn_ctx *c, long long B) {
  Plan p;
  p.fc_cap_floats = 0;
  long long off = 0;
  for (int l = 0; l < L_COUNT; ++l) {
    int N, K, bkn;
    layer_nk(c, l, &N, &K, &bkn);
    // capacity (offsets) from max_batch, split from the actual batch
    const int min_rows = l == L_HD ? 32 : 256;  // heads: one 4-row batch per wave of the fused heads + loss launch (8 waves)
    long long ms_cap = msplit_bound(layer_rows(c, l, c->max_batch), bkn, min_rows);
    pick_msplit(layer_rows(c, l, B), bkn, min_rows, &p.s[l].msplit, &p.s[l].mper);
    p.s[l].direct = 0;
    p.s[l].bsplit = 0;  // 0 = msplit (set below)
    long long b_cap = 0;  // bias partials beyond the weight slabs' count
    if (l == L_FC && tail_supported(c->flat, c->num_actions)) {  // the factored tail's partial G slabs live here
      // (tail_bwd_workgroups(B) is NOT monotone in B: the capacity is the bound over every B <= max_batch)
      const long long need = (tail_slab_capacity_floats(c->max_batch, c->num_actions) + static_cast<long long>(N) * K - 1) / (static_cast<long long>(N) * K);
      if (ms_cap < need) ms_cap = need;
    }
    if (l == L_FC && wgrad_direct_on()) {  // capacity for the dedicated kernel's slices / bias chunks
      const long long slices = fc_wgrad_slices(c->max_batch >= 256 ? c->max_batch : 256, K);
      if (ms_cap < slices) ms_cap = slices;
      b_cap = kBiasChunks;
      if (B >= wgrad_direct_min_batch() && fc_wgrad_supported(static_cast<int>(B), N, K)) {
        p.s[l].direct = 1;
        p.s[l].msplit = fc_wgrad_slices(static_cast<int>(B), K);
        p.s[l].bsplit = kBiasChunks;
      }
    }
    if (layer_direct_supported(c, l)) {
      const int st = l == L_C1 ? ST_CONV1_WGRAD : ST_CONV2_WGRAD;
      // (the bf16 x6 kernels of wgrad_b6.hip hold one image per CU: 256 workgroups; capacity for either family)
      const int nwg = wgrad_b6_on() ? 256 : wgrad_direct_workgroups(st, B);
      const int cap = wgrad_direct_workgroups(st, c->max_batch) > 256 ? wgrad_direct_workgroups(st, c->max_batch) : 256;
      if (ms_cap < cap) ms_cap = cap;
      if (B >= conv_wgrad_min_batch()) {
        p.s[l].direct = 1;
        p.s[l].msplit = static_cast<int>(B < nwg ? B : nwg);
      }
    }
    if (l == L_C0) {  // one slab per persistent workgroup of the direct conv0 wgrad (<= 512; the K-split kernel: <= 256)
      const long long tiles = (layer_rows(c, l, B) + 255) / 256;
      p.s[l].msplit = static_cast<int>(tiles < 512 ? tiles : 512);
      if (conv0_ks_usable(c)) p.s[l].msplit = conv0_wgrad_ks_workgroups(B);  // (never more than the tile kernel's count)
      p.s[l].mper = roundup((layer_rows(c, l, B) + p.s[l].msplit - 1) / p.s[l].msplit, 32);
    }
    if (p.s[l].bsplit == 0) p.s[l].bsplit = p.s[l].msplit;
    p.s[l].w_off = off;
    if (l == L_FC) p.fc_cap_floats = ms_cap * N * K;
    off += ms_cap * N * K;
    p.s[l].b_off = off;
    off += (b_cap > ms_cap ? b_cap : ms_cap) * N;
    off = (off + 63) / 64 * 64;
  }
  p.total = off;
  return p;
}

int dx_cnn_init(dx_cnn_ctx *c) {
  DX_TRACE("dx_cnn_init");
  DX_REQUIRE(c != nullptr, "dx_cnn_init: null ctx");
  DX_REQUIRE(c->struct_bytes == static_cast<int>(sizeof(dx_cnn_ctx)),
             "dx_cnn_init: struct size mismatch (caller %d, library %d)", c->struct_bytes,
             static_cast<int>(sizeof(dx_cnn_ctx)));
  DX_REQUIRE(c->in_c == 4, "dx_cnn_init: only 4 stacked frames are supported (in_c=%d)", c->in_c);
  DX_REQUIRE(c->in_h >= 36 && c->in_w >= 36, "dx_cnn_init: input %dx%d too small", c->in_h, c->in_w);
  DX_REQUIRE(c->num_actions >= 1 && c->num_actions <= 31, "dx_cnn_init: need 1 <= A <= 31");
  DX_REQUIRE(c->max_batch >= 1, "dx_cnn_init: max_batch < 1");
  const Derived d = derive(c);
  DX_REQUIRE(d.h2 >= 1 && d.w2 >= 1, "dx_cnn_init: input too small for the conv stack");
  c->h0 = d.h0; c->w0 = d.w0; c->h1 = d.h1; c->w1 = d.w1; c->h2 = d.h2; c->w2 = d.w2;
  c->flat = d.flat;
  const int A = c->num_actions;
  // canonical (state_dict) layout: weight, bias per layer in order
  const long long wsz[6] = {kC0 * c->in_c * 64LL, kC1 * kC0 * 16LL, kC2 * kC1 * 9LL,
                            static_cast<long long>(kHid) * d.flat, static_cast<long long>(A) * kHid, kHid};
  const long long bsz[6] = {kC0, kC1, kC2, kHid, A, 1};
  long long off = 0;
  for (int i = 0; i < 6; ++i) {
    c->off_w[i] = off; off += wsz[i];
    c->off_b[i] = off; off += bsz[i];
  }
  c->param_count = off;
  // packed mirrors
  long long po = 0;
  auto take = [&](long long n) { long long o = po; po += (n + 63) / 64 * 64; return o; };
  c->pk_c0f = take(wsz[0]); c->pk_c1f = take(wsz[1]); c->pk_c2f = take(wsz[2]); c->pk_fcf = take(wsz[3]);
  c->pk_hdf = take(kHeadLd * kHid); c->pk_hdb = take(kHeadLd);
  for (int p = 0; p < 4; ++p) c->pk_c1d[p] = take(kC0 * 4LL * kC1);
  c->pk_c2d = take(wsz[2]); c->pk_fcd = take(wsz[3]); c->pk_hdd = take(kHid * kHeadLd);
  // bf16 planes of the NT mirrors: 3 planes x 2 bytes = 1.5 floats per element
  auto take_planes = [&](long long n) { return take((3 * n + 1) / 2); };
  c->pb_c1f = take_planes(wsz[1]); c->pb_c2f = take_planes(wsz[2]); c->pb_fcf = take_planes(wsz[3]);
  c->pb_c1d = take_planes(wsz[1]); c->pb_c2d = take_planes(wsz[2]); c->pb_fcd = take_planes(wsz[3]);
  c->pb_c0f = take_planes(wsz[0]);
  c->pk_wc = take((tail_supported(d.flat, A) ? tail_rows(A) : 8LL) * d.flat); c->pk_beff = take(64);
  c->pk_wcs = take(tail_supported(d.flat, A) ? tail_pack_scratch_floats(A) : 0);
  c->ps_c1f = take(convstack_pack_elems(0) / 2); c->ps_c2f = take(convstack_pack_elems(1) / 2);
  c->ps_wc = take(convstack_pack_elems(2));
  c->ps_c1d = take(dgrad_b6_pack_elems(1) / 2); c->ps_c2d = take(dgrad_b6_pack_elems(2) / 2);
  c->packed_count = po;
  c->slab_count = make_plan(c, c->max_batch).total;
  const long long mb = c->max_batch;
  c->y0_count = mb * d.h0 * d.w0 * kC0;
  c->y1_count = mb * d.h1 * d.w1 * kC1;
  c->y2_count = mb * d.flat;
  c->hid_count = mb * kHid;
  c->head_count = mb * kHeadLd;
  // [parts][B][512] partial sums of the linear layer's forward: 7 parts up to 1,024 rows (rollout path,
  // small minibatches) and for the ring kernel's K split up to 2,560 rows (3,072 rows of capacity)
  long long hs = mb;
  const long long mid = mb < 3072 ? mb : 3072;
  if (hs < 7 * mid) hs = 7 * mid;
  // the rollout's weight-stationary linear layer (fc_rollout.hip) writes 14 parts, up to 1,024 rows
  const long long act_rows = mb < 1024 ? mb : 1024;
  if (hs < fc_rollout_parts() * act_rows) hs = fc_rollout_parts() * act_rows;
  c->hid_slab_count = hs * kHid;
  return DX_OK;
}

static int check_ctx(const dx_cnn_ctx *c, const char *who, long long B, bool need_bwd) {
  DX_REQUIRE(c && c->struct_bytes == static_cast<int>(sizeof(dx_cnn_ctx)) && c->param_count > 0,
             "%s: ctx not initialised by dx_cnn_init", who);
  DX_REQUIRE(B >= 1 && B <= c->max_batch, "%s: batch %lld outside [1, max_batch=%d]", who, B, c->max_batch);
  DX_REQUIRE(c->params && c->packed && c->y0 && c->y1 && c->y2 && c->hid && c->head,
             "%s: forward buffers not set", who);
  if (need_bwd)
    DX_REQUIRE(c->grads && c->dy0 && c->dy1 && c->dy2 && c->dhid && c->dhead && c->slabs,
               "%s: backward buffers not set", who);
  return DX_OK;
}

static uint16_t *planes(const dx_cnn_ctx *c, long long off) {
  return reinterpret_cast<uint16_t *>(c->packed + off);
}

// canonical parameters -> packed mirrors.  part 1 = what the FIRST conv layer's forward reads (its
// forward mirror and the bf16 planes of it), part 2 = every other mirror, 3 = both.  The split lets
// dx_cnn_ppo_epoch start the next minibatch's first layer while the rest is still being packed.
static bool convstack_train_env() {
  return DX_ENV("DX_CONVSTACK_TRAIN", 1) != 0;
}
static bool dgrad_b6_usable(const dx_cnn_ctx *c) {  // (the kernels are built for the conv stack of an 84 x 84 observation)
  return dgrad_b6_on() && c->h0 == 20 && c->w0 == 20 && c->h1 == 9 && c->w1 == 9 && c->h2 == 7 && c->w2 == 7;
}
static bool fc_factored(const dx_cnn_ctx *c) { return fc_factored_env() && tail_supported(c->flat, c->num_actions); }
// DX_TAIL_FUSED=0: the factored tail's loss (tail_loss_kernel) and its backward pass over y2 (tail_bwd_kernel) as two
// launches; default (up to 7 actions): ONE launch from dx_cnn_heads_loss_f32 (heads.hip: tail_loss_bwd_kernel), and the
// backward that follows it (dx_cnn_backward_part 2 / 3) starts at conv2
static bool tail_fused(const dx_cnn_ctx *c) { return fc_factored(c) && c->num_actions + 1 <= 8 && DX_ENV("DX_TAIL_FUSED", 1) != 0; }

// `light`: without the mirrors only the layer-by-layer linear layer / heads read (pk_fcf, pk_fcd and their
// planes): what dx_cnn_ppo_epoch packs between the updates of an epoch when the factored tail is on
static int pack_part(const dx_cnn_ctx *c, int part, hipStream_t s, bool light = false, bool direct = false) {
  const int A = c->num_actions, IC0 = c->in_c, P = c->h2 * c->w2, flat = c->flat;
  const float *w = c->params;
  float *pk = c->packed;
  float *wcf = convstack_supported(c->in_h, c->in_w, c->in_c) ? pk + c->ps_wc : nullptr;  // Wc in the conv-stack kernel's order
  if (direct) {
    // between two updates of an epoch on the default route nothing reads an fp32 mirror: Wc, and the bf16 planes
    // of the three conv layers in the orders their kernels read, straight from the parameters (pack_direct_dev.hpp)
    // (two launches: the conv planes ride in extra workgroups of the tail pack's first kernel)
    const TailDirectPlanes direct_planes{planes(c, c->pb_c0f), planes(c, c->ps_c1f), planes(c, c->ps_c2f), planes(c, c->ps_c1d),
                                         planes(c, c->ps_c2d)};
    return launch_tail_pack(w, c->off_w, c->off_b, A, pk + c->pk_wc, pk + c->pk_beff, pk + c->pk_wcs, wcf, s, &direct_planes);
  }
  // The padded head rows / columns beyond A + 1 are never written: `packed` must be zero-filled
  // once by its owner (include/derl_amd.h), not on every parameter update.
  PermuteJob j[kMaxJobs];
  int n = 0;
  auto add = [&](const float *src, float *dst, long long total, int D1, int D2, int D3, long long s0,
                 long long s1, long long s2, long long s3, long long off) {
    j[n++] = PermuteJob{src, dst, total, D1, D2, D3, s0, s1, s2, s3, off, 1, 0, 0};
  };
  const long long wsz0 = kC0 * 64LL * IC0, wsz1 = kC1 * 16LL * kC0, wsz2 = kC2 * 9LL * kC1,
                  wsz3 = static_cast<long long>(kHid) * flat;
  // conv forward: dst [oc][kh][kw][ic] <- OIHW
  if (part & 1) add(w + c->off_w[0], pk + c->pk_c0f, wsz0, 8, 8, IC0, IC0 * 64LL, 8, 1, 64, 0);
  if (part == 1) {  // one-slab "reduction" = the plain permutation, then the planes of the mirror
    if (int rc = launch_permute_reduce(j, n, s)) return rc;
    const float *src[1] = {pk + c->pk_c0f};
    uint16_t *dst[1] = {planes(c, c->pb_c0f)};
    const long long cnt[1] = {wsz0};
    return launch_split_planes(src, dst, cnt, 1, s);
  }
  add(w + c->off_w[1], pk + c->pk_c1f, wsz1, 4, 4, kC0, kC0 * 16LL, 4, 1, 16, 0);
  add(w + c->off_w[2], pk + c->pk_c2f, wsz2, 3, 3, kC1, kC1 * 9LL, 3, 1, 9, 0);
  // linear layer: launch_fc_pack below (dst [n][p][c] <- [n][c*P + p], and its transpose)
  // heads: rows 0..A-1 policy logits, row A value
  add(w + c->off_w[4], pk + c->pk_hdf, static_cast<long long>(A) * kHid, 1, 1, kHid, kHid, 0, 0, 1, 0);
  add(w + c->off_w[5], pk + c->pk_hdf + static_cast<long long>(A) * kHid, kHid, 1, 1, kHid, kHid, 0, 0, 1, 0);
  add(w + c->off_b[4], pk + c->pk_hdb, A, 1, 1, 1, 1, 0, 0, 0, 0);
  add(w + c->off_b[5], pk + c->pk_hdb + A, 1, 1, 1, 1, 1, 0, 0, 0, 0);
  // conv1 dgrad, one pack per input-pixel parity (py,px): dst [ic][a][b][oc] <- W[oc][ic][py+2a][px+2b]
  for (int p = 0; p < 4; ++p)
    add(w + c->off_w[1], pk + c->pk_c1d[p], kC0 * 4LL * kC1, 2, 2, kC1, 16, 8, 2, kC0 * 16LL,
        (p >> 1) * 4 + (p & 1));
  // conv2 dgrad: dst [ic][kh][kw][oc]
  add(w + c->off_w[2], pk + c->pk_c2d, kC1 * 9LL * kC2, 3, 3, kC2, 9, 3, 1, kC1 * 9LL, 0);
  // heads dgrad: dst [k][32] <- heads [j][k] (scatter: the source rows are the contiguous side)
  add(w + c->off_w[4], pk + c->pk_hdd, static_cast<long long>(A) * kHid, kHid, 1, 1, 1, kHeadLd, 0, 0, 0);
  j[n - 1].scatter = 1;
  add(w + c->off_w[5], pk + c->pk_hdd + A, kHid, kHid, 1, 1, 1, kHeadLd, 0, 0, 0);
  j[n - 1].scatter = 1;
  if (fc_factored(c))  // Wc = Wh Wfc and beff: the factored tail's only mirror
    if (int rc = launch_tail_pack(w, c->off_w, c->off_b, A, pk + c->pk_wc, pk + c->pk_beff, pk + c->pk_wcs, wcf, s)) return rc;
  if (light) {
    if (int rc = launch_permute_reduce(j, n, s)) return rc;
  } else if (int rc = launch_pack_fused(j, n, w + c->off_w[3], pk + c->pk_fcf, pk + c->pk_fcd, kHid, P, kC2, s)) {
    return rc;
  }
  // bf16 planes (second launch: reads the mirrors packed above): the three conv layers' for the rollout's
  // conv-stack kernel, the other NT mirrors only for the opt-in bf16-split GEMMs
  const float *src[7] = {pk + c->pk_c0f, pk + c->pk_c1f, pk + c->pk_c2f, pk + c->pk_fcf, pk + c->pk_c1d[0],
                         pk + c->pk_c2d, pk + c->pk_fcd};
  uint16_t *dst[7] = {planes(c, c->pb_c0f), planes(c, c->pb_c1f), planes(c, c->pb_c2f), planes(c, c->pb_fcf),
                      planes(c, c->pb_c1d), planes(c, c->pb_c2d), planes(c, c->pb_fcd)};
  const long long cnt[7] = {wsz0, wsz1, wsz2, wsz3, wsz1, wsz2, wsz3};
  // (conv0 / conv1 / conv2: operands of the rollout's one-launch conv stack, convstack.hip)
  const int first = (part & 1) ? 0 : 1, count = (use_b3() ? 7 : 3) - first;
  if (int rc = launch_split_planes(src + first, dst + first, cnt + first, count, s)) return rc;
  if (dgrad_b6_usable(c)) {  // the data-gradient kernels' fragment-order planes, from the fp32 mirrors packed above
    const float *c1d[4] = {pk + c->pk_c1d[0], pk + c->pk_c1d[1], pk + c->pk_c1d[2], pk + c->pk_c1d[3]};
    if (int rc = launch_dgrad_b6_pack(c1d, pk + c->pk_c2d, planes(c, c->ps_c1d), planes(c, c->ps_c2d), s)) return rc;
  }
  // the conv-stack kernel's fragment-order copies (only the rollout reads them: not between an epoch's updates)
  if ((!light || convstack_train_env()) && convstack_supported(c->in_h, c->in_w, c->in_c))
    return launch_convstack_pack(planes(c, c->pb_c1f), planes(c, c->pb_c2f), nullptr,  // (Wc's copy: launch_tail_pack above)
                                 planes(c, c->ps_c1f), planes(c, c->ps_c2f), pk + c->ps_wc, s);
  return DX_OK;
}

// canonical parameters -> packed mirrors (call after every parameter change)
int dx_cnn_pack(const dx_cnn_ctx *c, void *stream) {
  DX_TRACE("dx_cnn_pack");
  if (int rc = check_ctx(c, "dx_cnn_pack", 1, false)) return rc;
  return pack_part(c, 3, as_stream(stream));
}

static Gather conv_gather(const void *src, const int32_t *idx, int H, int W, int C, int OH, int OW,
                          int stride, int KH, int KW) {
  Gather g;
  std::memset(&g, 0, sizeof(g));
  g.src = src; g.idx = idx;
  g.img_stride = static_cast<long long>(H) * W * C;
  g.H = H; g.W = W; g.C = C;
  g.OHW = OH * OW; g.OW = OW;
  g.div_img = make_fastdiv(g.OHW); g.div_row = make_fastdiv(OW);
  g.sy = g.sx = stride;
  g.nseg = KH; g.seglen = KW * C; g.check = 0;
  for (int kh = 0; kh < KH; ++kh) { g.seg_off[kh] = kh * W * C; g.seg_dy[kh] = kh; g.seg_dx[kh] = 0; }
  return g;
}

static Gather rows_gather(const void *src, int width) {  // plain [rows][width] matrix
  Gather g;
  std::memset(&g, 0, sizeof(g));
  g.src = src; g.img_stride = width; g.H = g.W = 1; g.C = width;
  g.OHW = 1; g.OW = 1; g.div_img = make_fastdiv(1); g.div_row = make_fastdiv(1);
  g.sy = g.sx = 1; g.nseg = 1; g.seglen = width;
  return g;
}

// dgrad rows = pixels (y', x') of one parity class of the conv input; taps (a, b) read the
// output-gradient pixel (y'-a, x'-b)
static Gather dgrad_gather(const void *dy, int H, int W, int C, int OH, int OW, int TA, int TB) {
  Gather g;
  std::memset(&g, 0, sizeof(g));
  g.src = dy; g.img_stride = static_cast<long long>(H) * W * C;
  g.H = H; g.W = W; g.C = C;
  g.OHW = OH * OW; g.OW = OW;
  g.div_img = make_fastdiv(g.OHW); g.div_row = make_fastdiv(OW);
  g.sy = g.sx = 1; g.nseg = TA * TB; g.seglen = C; g.check = 1;
  for (int a = 0; a < TA; ++a)
    for (int b = 0; b < TB; ++b) {
      const int s = a * TB + b;
      g.seg_off[s] = (-a * W - b) * C; g.seg_dy[s] = -a; g.seg_dx[s] = -b;
    }
  return g;
}

static NTArgs nt_args(const Gather &g, const float *Wp, const float *bias, float *out, int ldc,
                      long long M, int N, int K) {
  NTArgs a;
  std::memset(&a, 0, sizeof(a));
  a.g = g; a.Wp = Wp; a.bias = bias; a.out = out; a.ldc = ldc;
  a.M = static_cast<int>(M); a.N = N; a.K = K; a.ksplit = 1;
  return a;
}

// slabs -> canonical gradients.  `which`: bit 0 = the three conv layers, bit 1 = linear layer +
// heads (the contiguous tail of the flat gradient buffer: its all-reduce can start while the conv
// layers' backward still runs)
static int finalize_grads(const dx_cnn_ctx *c, const Plan &plan, int which, hipStream_t s, const TailGreduceArgs *greduce = nullptr) {
  const int IC0 = c->in_c, A = c->num_actions, flat = c->flat, P = c->h2 * c->w2;
  PermuteJob j[kMaxJobs];
  int n = 0;
  float *g = c->grads;
  auto addw = [&](int layer, long long off_dst, long long total, int D1, int D2, int D3, long long s0,
                  long long s1, long long s2, long long s3, long long off, int N, int K, int scatter) {
    j[n++] = PermuteJob{c->slabs + plan.s[layer].w_off, g + off_dst, total, D1, D2, D3, s0, s1, s2, s3,
                        off, plan.s[layer].msplit, static_cast<long long>(N) * K, scatter};
  };
  auto addb = [&](int layer, long long off_dst, long long total, long long off, int N) {
    j[n++] = PermuteJob{c->slabs + plan.s[layer].b_off, g + off_dst, total, 1, 1, 1, 1, 0, 0, 0, off,
                        plan.s[layer].bsplit, N, 0};
  };
  // which: 1 = the three conv layers (= 4 | 8), 2 = linear layer + heads, 4 = conv0 only, 8 = conv1 + conv2
  if (which & 1) which |= 4 | 8;
  // conv: iterate the slab [oc][kh][kw][ic] in its own order (coalesced slab reads) and
  // scatter into the canonical (oc, ic, kh, kw) layout
  if (which & 4) {
    addw(L_C0, c->off_w[0], kC0 * 64LL * IC0, 8, 8, IC0, IC0 * 64LL, 8, 1, 64, 0, kC0, 64 * IC0, 1);
    addb(L_C0, c->off_b[0], kC0, 0, kC0);
  }
  if (which & 8) {
    addw(L_C1, c->off_w[1], kC1 * 16LL * kC0, 4, 4, kC0, kC0 * 16LL, 4, 1, 16, 0, kC1, 16 * kC0, 1);
    addw(L_C2, c->off_w[2], kC2 * 9LL * kC1, 3, 3, kC1, kC1 * 9LL, 3, 1, 9, 0, kC2, 9 * kC1, 1);
    addb(L_C1, c->off_b[1], kC1, 0, kC1);
    addb(L_C2, c->off_b[2], kC2, 0, kC2);
  }
  if (which & 2) {
    // linear weight: launch_fc_grad_finalize below (slab [n][p][c] -> canonical [n][c*P + p])
    addw(L_HD, c->off_w[4], static_cast<long long>(A) * kHid, kHid, 1, 1, kHid, 1, 0, 0, 0, kHeadLd, kHid, 0);
    addw(L_HD, c->off_w[5], kHid, kHid, 1, 1, kHid, 1, 0, 0, static_cast<long long>(A) * kHid, kHeadLd, kHid, 0);
    addb(L_FC, c->off_b[3], kHid, 0, kHid);
    addb(L_HD, c->off_b[4], A, 0, kHeadLd);
    addb(L_HD, c->off_b[5], 1, A, kHeadLd);
    // one launch for the permute jobs and the linear layer's [p][c] -> [c][p] reduction
    return launch_finalize_fused(j, n, c->slabs + plan.s[L_FC].w_off, plan.s[L_FC].msplit,
                                 static_cast<long long>(kHid) * flat, g + c->off_w[3], kHid, P, kC2, s);
  }
  if (greduce != nullptr) return launch_permute_reduce_greduce(j, n, *greduce, s);
  return launch_permute_reduce(j, n, s);
}

// One launch (one profiler row) of the network.  Forward = stages ST_CONV0_FWD..ST_HEADS_FWD,
// backward = ST_HEADS_WGRAD..ST_FINALIZE in enum order.
static int run_stage(const dx_cnn_ctx *c, int stage, const void *obs, int obs_is_u8,
                     const int32_t *sample_idx, int B, const Plan &plan, hipStream_t s) {
  const float *w = c->params;
  const float *pk = c->packed;
  const int IC0 = c->in_c, flat = c->flat, P = c->h2 * c->w2;
  const long long M0 = static_cast<long long>(B) * c->h0 * c->w0, M1 = static_cast<long long>(B) * c->h1 * c->w1,
                  M2 = static_cast<long long>(B) * P;
  auto tn = [&](int layer, const Gather &g, const float *G, int ldg, long long M, int N, int K, bool u8) {
    TNArgs t;
    std::memset(&t, 0, sizeof(t));
    t.g = g; t.G = G; t.ldg = ldg;
    t.slab = c->slabs + plan.s[layer].w_off; t.bias_slab = c->slabs + plan.s[layer].b_off;
    t.M = static_cast<int>(M); t.N = N; t.K = K;
    t.msplit = plan.s[layer].msplit; t.mper = plan.s[layer].mper;
    return launch_tn(t, u8, stage, s);
  };
  // which kernel family ran this stage (dx_cnn_last_route: the perf guard against a batch silently
  // leaving the ring / direct kernels, and what bench.py prints per stage)
  auto took = [&](const char *family, int rc) {
    if (rc != DX_ENOSUP && stage >= 0 && stage < ST_COUNT) g_route[stage] = family;
    return rc;
  };
  NTArgs a;
  switch (stage) {
    case ST_CONV0_FWD:
      a = nt_args(conv_gather(obs, sample_idx, c->in_h, c->in_w, IC0, c->h0, c->w0, 4, 8, 8), pk + c->pk_c0f,
                  w + c->off_b[0], c->y0, kC0, M0, kC0, 64 * IC0);
      if (obs_is_u8 && IC0 == 4 && M0 <= 32LL * conv0_lat_max_tiles() && !conv0_f32()) {  // rollout-sized batches
        Conv0Args d;
        std::memset(&d, 0, sizeof(d));
        d.obs = static_cast<const uint8_t *>(obs); d.idx = sample_idx;
        d.in_h = c->in_h; d.in_w = c->in_w; d.h0 = c->h0; d.w0 = c->w0;
        d.M = static_cast<int>(M0); d.bias = w + c->off_b[0]; d.out = c->y0;
        const int rc = took("conv0_lat_b16", launch_conv0_lat_b16(d, planes(c, c->pb_c0f), s));
        if (rc != DX_ENOSUP) return rc;
      }
      if (obs_is_u8 && conv0_f32() && M0 <= 32LL * conv0_lat_max_tiles()) {  // fp32-MFMA latency kernel
        const int rc = took("igemm_lat", launch_nt_lat(a, true, EPI_BIAS_RELU, stage, s));
        if (rc != DX_ENOSUP) return rc;
      }
      if (obs_is_u8 && conv0_direct_supported(c->in_h, c->in_w, IC0, c->h0, c->w0)) {
        Conv0Args d;
        std::memset(&d, 0, sizeof(d));
        d.obs = static_cast<const uint8_t *>(obs); d.idx = sample_idx;
        d.in_h = c->in_h; d.in_w = c->in_w; d.h0 = c->h0; d.w0 = c->w0;
        d.M = static_cast<int>(M0); d.ntiles = static_cast<int>((M0 + 255) / 256);
        d.Wp = pk + c->pk_c0f; d.bias = w + c->off_b[0]; d.out = c->y0;
        return conv0_f32() ? took("conv0_f32", launch_conv0_fwd(d, s)) : took("conv0_b16", launch_conv0_fwd_b16(d, planes(c, c->pb_c0f), s));
      }
      return took("igemm_nt", launch_nt(a, obs_is_u8 != 0, EPI_BIAS_RELU, stage, s));
    case ST_CONV1_FWD:
      a = nt_args(conv_gather(c->y0, nullptr, c->h0, c->w0, kC0, c->h1, c->w1, 2, 4, 4), pk + c->pk_c1f,
                  w + c->off_b[1], c->y1, kC1, M1, kC1, 16 * kC0);
      a.Wb = planes(c, c->pb_c1f); a.wb_plane = kC1 * 16LL * kC0;
      if (const int rc = took("ntp", launch_ntp_fwd(a, stage, s)); rc != DX_ENOSUP) return rc;
      return took("igemm_nt", launch_nt(a, false, EPI_BIAS_RELU, stage, s));
    case ST_CONV2_FWD:
      a = nt_args(conv_gather(c->y1, nullptr, c->h1, c->w1, kC1, c->h2, c->w2, 1, 3, 3), pk + c->pk_c2f,
                  w + c->off_b[2], c->y2, kC2, M2, kC2, 9 * kC1);
      a.Wb = planes(c, c->pb_c2f); a.wb_plane = kC2 * 9LL * kC1;
      if (const int rc = took("ntp", launch_ntp_fwd(a, stage, s)); rc != DX_ENOSUP) return rc;
      return took("igemm_nt", launch_nt(a, false, EPI_BIAS_RELU, stage, s));
    case ST_FC_FWD: {
      // small minibatches (multi-GPU shards): 49 sequential 64-deep K steps on 128 workgroups are
      // a latency chain; split K 7 ways into the rollout slabs and sum them into hid
      // (the ring kernel first: from 1,024 rows up it covers the layer, with K in 7 / 2 parts through the
      // same slabs while its tiles are few)
      if (B >= 1024 && ntp_fc_fwd()) {
        if (const int rc = took("ntp", launch_ntp_rows(c->y2, flat, pk + c->pk_fcf, nullptr, w + c->off_b[3], c->hid, B, kHid, flat, c->hid_slabs, c->hid_slabs ? c->hid_slab_count : 0, s));
            rc != DX_ENOSUP)
          return rc;
      }
      const int ks = (B <= 1024 && c->hid_slabs && 7LL * B * kHid <= c->hid_slab_count && flat % (7 * 64) == 0) ? 7 : 1;
      a = nt_args(rows_gather(c->y2, flat), pk + c->pk_fcf, w + c->off_b[3], ks > 1 ? c->hid_slabs : c->hid,
                  kHid, B, kHid, flat);
      a.Wb = planes(c, c->pb_fcf); a.wb_plane = static_cast<long long>(kHid) * flat;
      a.ksplit = ks;
      a.slab_stride = static_cast<long long>(B) * kHid;
      if (ks == 1 && nt_dma_on() && nt_dma_supported(B, kHid, flat)) {
        const NtDmaArgs d{c->y2, pk + c->pk_fcf, w + c->off_b[3], nullptr, c->hid, B, kHid, flat, flat, kHid};
        return took("nt_dma", launch_nt_dma(d, EPI_BIAS, s));
      }
      if (int rc = took(ks > 1 ? "igemm_nt split-K" : "igemm_nt", launch_nt(a, false, EPI_BIAS, stage, s))) return rc;
      if (ks == 1) return DX_OK;
      PermuteJob jr{c->hid_slabs, c->hid, static_cast<long long>(B) * kHid, 1, 1, 1, 1, 0, 0, 0, 0, ks,
                    static_cast<long long>(B) * kHid, 0};
      return launch_permute_reduce(&jr, 1, s);
    }
    case ST_HEADS_FWD:
      a = nt_args(rows_gather(c->hid, kHid), pk + c->pk_hdf, pk + c->pk_hdb, c->head, kHeadLd, B, kHeadLd, kHid);
      return took("igemm_nt", launch_nt(a, false, EPI_BIAS, stage, s));
    case ST_HEADS_WGRAD:
      return took("igemm_tn", tn(L_HD, rows_gather(c->hid, kHid), c->dhead, kHeadLd, B, kHeadLd, kHid, false));
    case ST_HEADS_DGRAD:
      a = nt_args(rows_gather(c->dhead, kHeadLd), pk + c->pk_hdd, nullptr, c->dhid, kHid, B, kHid, kHeadLd);
      return took("igemm_nt", launch_nt(a, false, EPI_NONE, stage, s));
    case ST_FC_WGRAD:
      if (plan.s[L_FC].direct) {
        if (int rc = launch_colsum(c->dhid, c->slabs + plan.s[L_FC].b_off, B, kHid, plan.s[L_FC].bsplit, s)) return rc;
        const FcWgradArgs d{c->dhid, c->y2, c->slabs + plan.s[L_FC].w_off, B, flat, plan.s[L_FC].msplit, 0, 0};
        return took("wgrad_fc", launch_fc_wgrad(d, s));
      }
      return took("igemm_tn", tn(L_FC, rows_gather(c->y2, flat), c->dhid, kHid, B, kHid, flat, false));
    case ST_FC_DGRAD:
      if (const int rc = took("ntp", launch_ntp_rows(c->dhid, kHid, pk + c->pk_fcd, c->y2, nullptr, c->dy2, B, flat, kHid, nullptr, 0, s)); rc != DX_ENOSUP)
        return rc;
      if (nt_dma_on() && nt_dma_supported(B, flat, kHid)) {
        const NtDmaArgs d{c->dhid, pk + c->pk_fcd, nullptr, c->y2, c->dy2, B, flat, kHid, kHid, flat};
        return took("nt_dma", launch_nt_dma(d, EPI_MASK, s));
      }
      a = nt_args(rows_gather(c->dhid, kHid), pk + c->pk_fcd, nullptr, c->dy2, flat, B, flat, kHid);
      a.Wb = planes(c, c->pb_fcd); a.wb_plane = static_cast<long long>(kHid) * flat;
      a.mask_src = c->y2;
      return took("igemm_nt", launch_nt(a, false, EPI_MASK, stage, s));
    case ST_CONV2_WGRAD:
      if (plan.s[L_C2].direct) {
        const WgradDirectArgs d{c->y1, c->dy2, c->slabs + plan.s[L_C2].w_off, c->slabs + plan.s[L_C2].b_off,
                                B, c->h1, c->w1, c->h2, c->w2, 0, bwd_descending(ST_CONV2_WGRAD) ? 1 : 0};
        if (wgrad_b6_on()) return took("wgrad_b6", launch_wgrad_b6(d, stage, plan.s[L_C2].msplit, s));
        return took("wgrad_direct", launch_wgrad_direct(d, stage, plan.s[L_C2].msplit, s));
      }
      return took("igemm_tn", tn(L_C2, conv_gather(c->y1, nullptr, c->h1, c->w1, kC1, c->h2, c->w2, 1, 3, 3), c->dy2, kC2, M2,
                                  kC2, 9 * kC1, false));
    case ST_CONV2_DGRAD:
      if (dgrad_b6_usable(c)) return took("dgrad_b6", launch_dgrad_b6(2, c->dy2, planes(c, c->ps_c2d), c->y1, c->dy1, B, s, bwd_descending(ST_CONV2_DGRAD)));
      a = nt_args(dgrad_gather(c->dy2, c->h2, c->w2, kC2, c->h1, c->w1, 3, 3), pk + c->pk_c2d, nullptr,
                  c->dy1, kC1, M1, kC1, 9 * kC2);
      a.Wb = planes(c, c->pb_c2d); a.wb_plane = kC2 * 9LL * kC1;
      a.mask_src = c->y1;
      if (const int rc = took("ntp", launch_ntp_pix(a, B, 3, 3, s)); rc != DX_ENOSUP) return rc;
      if (const int rc = took("igemm_pix", launch_nt_pix(a, B, EPI_MASK, stage, s)); rc != DX_ENOSUP) return rc;
      return took("igemm_nt", launch_nt(a, false, EPI_MASK, stage, s));
    case ST_CONV1_WGRAD:
      if (plan.s[L_C1].direct) {
        const WgradDirectArgs d{c->y0, c->dy1, c->slabs + plan.s[L_C1].w_off, c->slabs + plan.s[L_C1].b_off,
                                B, c->h0, c->w0, c->h1, c->w1, 0, bwd_descending(ST_CONV1_WGRAD) ? 1 : 0};
        if (wgrad_b6_on()) return took("wgrad_b6", launch_wgrad_b6(d, stage, plan.s[L_C1].msplit, s));
        return took("wgrad_direct", launch_wgrad_direct(d, stage, plan.s[L_C1].msplit, s));
      }
      return took("igemm_tn", tn(L_C1, conv_gather(c->y0, nullptr, c->h0, c->w0, kC0, c->h1, c->w1, 2, 4, 4), c->dy1, kC1, M1,
                                  kC1, 16 * kC0, false));
    case ST_CONV1_DGRAD: {
      // 4x4 stride-2 conv: input pixel (2y'+py, 2x'+px) receives taps kh = py+2a, kw = px+2b from
      // output-gradient pixel (y'-a, x'-b) for every parity (py,px), so the four parity classes
      // share one gathered A matrix: one GEMM with N = 4 x 32 columns [(py,px)][ic], scattered
      // to the 2x2 pixel block by the output map.
      const int OHp = (c->h0 + 1) / 2, OWp = (c->w0 + 1) / 2;
      if (dgrad_b6_usable(c)) return took("dgrad_b6", launch_dgrad_b6(1, c->dy1, planes(c, c->ps_c1d), c->y0, c->dy0, B, s, bwd_descending(ST_CONV1_DGRAD)));
      a = nt_args(dgrad_gather(c->dy1, c->h1, c->w1, kC1, OHp, OWp, 2, 2), pk + c->pk_c1d[0], nullptr,
                  c->dy0, kC0, static_cast<long long>(B) * OHp * OWp, 4 * kC0, 4 * kC1);
      a.Wb = planes(c, c->pb_c1d); a.wb_plane = kC1 * 16LL * kC0;
      a.mask_src = c->y0;
      a.om.enabled = 1;
      a.om.OHW = OHp * OWp; a.om.OW = OWp;
      a.om.div_img = make_fastdiv(a.om.OHW); a.om.div_row = make_fastdiv(OWp);
      a.om.OUT_H = c->h0; a.om.OUT_W = c->w0; a.om.osy = a.om.osx = 2; a.om.chan = kC0;
      if (const int rc = took("ntp", launch_ntp_pix(a, B, 2, 2, s)); rc != DX_ENOSUP) return rc;
      if (const int rc = took("igemm_pix", launch_nt_pix(a, B, EPI_MASK, stage, s)); rc != DX_ENOSUP) return rc;
      return took("igemm_nt", launch_nt(a, false, EPI_MASK, stage, s));
    }
    case ST_CONV0_WGRAD:
      if (obs_is_u8 && conv0_direct_supported(c->in_h, c->in_w, IC0, c->h0, c->w0)) {
        Conv0Args d;
        std::memset(&d, 0, sizeof(d));
        d.obs = static_cast<const uint8_t *>(obs); d.idx = sample_idx;
        d.in_h = c->in_h; d.in_w = c->in_w; d.h0 = c->h0; d.w0 = c->w0;
        d.M = static_cast<int>(M0); d.ntiles = static_cast<int>((M0 + 255) / 256);
        d.G = c->dy0; d.slab = c->slabs + plan.s[L_C0].w_off; d.bias_slab = c->slabs + plan.s[L_C0].b_off;
        if (conv0_ks_usable(c)) return took("conv0_ks", launch_conv0_wgrad_ks(d, plan.s[L_C0].msplit, s));
        return conv0_f32() ? took("conv0_f32", launch_conv0_wgrad(d, plan.s[L_C0].msplit, s))
                           : took("conv0_b16", launch_conv0_wgrad_b16(d, plan.s[L_C0].msplit, s));
      }
      return took("igemm_tn", tn(L_C0, conv_gather(obs, sample_idx, c->in_h, c->in_w, IC0, c->h0, c->w0, 4, 8, 8), c->dy0, kC0,
                                  M0, kC0, 64 * IC0, obs_is_u8 != 0));
    case ST_FINALIZE:  // (with the factored tail the linear layer / heads have no slabs: the conv layers only)
      return took("finalize", finalize_grads(c, plan, fc_factored(c) ? 1 : 3, s));
    default:
      return fail(DX_EINVAL, "dx_cnn: unknown stage %d", stage);
  }
}

// DX_FWD_LANES=1 (an experiment, off by default): the forward as two independent chains over the two
// halves of the rows, the second on the side stream -- what pays for the rollout (two lanes) and
// for the backward (weight gradients beside data gradients) does NOT pay here: measured per PPO
// iteration, same box, alternating, 256 envs 41.9 -> 42.8 ms, 128 envs 24.4 -> 25.6 ms.  The two
// chains run the SAME stage at the same time and compete for the same weights in L2 and the same
// workgroup slots (each stage already fills 512 of them); half-sized launches only add their ramps.
// Rows are independent in every forward stage, so the activations are those of the one-chain forward
// (tests/test_cnn_gpu.py passes with the switch on).
static int fc_ksplit(int B, int flat);
static bool fwd_lanes(int B) {
  const int mode = DX_ENV("DX_FWD_LANES", 0), limit = DX_ENV("DX_FWD_LANE_MIN", 4096);
  return B >= 2 && B % 2 == 0 && (mode == 1 || (mode == -1 && B >= limit));
}

// The conv stack of a training minibatch as ONE launch of the image-resident kernel (convstack.hip, `train`):
// all three layers on the bf16 matrix cores, y0 / y1 / y2 stored for the backward.  DX_CONVSTACK_TRAIN=0: the
// three layer-by-layer stages.
static bool convstack_train_usable(const dx_cnn_ctx *c, int obs_is_u8) {
  return convstack_train_env() && obs_is_u8 && !conv0_f32() && convstack_supported(c->in_h, c->in_w, c->in_c);
}
static ConvStackArgs convstack_args(const dx_cnn_ctx *c, uint8_t *obs, int B);
static int convstack_train_forward(const dx_cnn_ctx *c, const void *obs, const int32_t *sample_idx, int B, hipStream_t s) {
  ConvStackArgs cs = convstack_args(c, static_cast<uint8_t *>(const_cast<void *>(obs)), B);
  cs.train = 1; cs.sample_idx = sample_idx;
  cs.y0 = c->y0; cs.y1 = c->y1; cs.y2 = c->y2;
  if (int rc = launch_convstack(cs, s)) return rc;
  g_route[ST_CONV0_FWD] = g_route[ST_CONV1_FWD] = g_route[ST_CONV2_FWD] = "convstack_train";
  return DX_OK;
}

// forward stages first .. last on `s` (as one chain, or as two half-batch chains joined at the end)
static int forward_stages(const dx_cnn_ctx *c, int first, int last, const void *obs, int obs_is_u8,
                          const int32_t *sample_idx, int B, hipStream_t s) {
  if (first == ST_CONV0_FWD && last >= ST_CONV2_FWD && convstack_train_usable(c, obs_is_u8)) {
    if (int rc = convstack_train_forward(c, obs, sample_idx, B, s)) return rc;
    first = ST_CONV2_FWD + 1;
    if (first > last) return DX_OK;
  }
  SideStream *side = fwd_lanes(B) ? side_stream() : nullptr;
  if (side == nullptr) {
    const Plan plan = make_plan(c, B);
    for (int st = first; st <= last; ++st)
      if (int rc = run_stage(c, st, obs, obs_is_u8, sample_idx, B, plan, s)) return rc;
    return DX_OK;
  }
  const int part = B / 2;
  dx_cnn_ctx upper = *c;  // rows part .. B-1
  upper.y0 += static_cast<long long>(part) * c->h0 * c->w0 * kC0;
  upper.y1 += static_cast<long long>(part) * c->h1 * c->w1 * kC1;
  upper.y2 += static_cast<long long>(part) * c->flat;
  upper.hid += static_cast<long long>(part) * kHid;
  upper.head += static_cast<long long>(part) * kHeadLd;
  dx_cnn_ctx lower = *c;  // rows 0 .. part-1: the same buffers, but only its half of the split-K scratch
  if (upper.hid_slabs) {  // the linear layer's split-K partials of the two halves side by side
    const long long half = c->hid_slab_count / 2 / 4 * 4;
    upper.hid_slabs += half;
    upper.hid_slab_count -= half;
    lower.hid_slab_count = half;
  }
  const void *obs_upper = obs;
  const int32_t *idx_upper = nullptr;
  if (sample_idx) idx_upper = sample_idx + part;
  else obs_upper = static_cast<const char *>(obs) + static_cast<long long>(part) * c->in_h * c->in_w * c->in_c * (obs_is_u8 ? 1 : 4);
  const Plan plan = make_plan(c, part);
  if (hipEventRecord(side->fork, s) != hipSuccess || hipStreamWaitEvent(side->stream[0], side->fork, 0) != hipSuccess)
    return fail(DX_EHIP, "forward: cannot order the side stream");
  int rc = DX_OK;
  for (int st = first; st <= last && rc == DX_OK; ++st) {
    rc = run_stage(&lower, st, obs, obs_is_u8, sample_idx, part, plan, s);
    if (rc == DX_OK) rc = run_stage(&upper, st, obs_upper, obs_is_u8, idx_upper, part, plan, side->stream[0]);
  }
  // also after a failed launch: the caller's stream never runs ahead of the side stream
  const bool joined = hipEventRecord(side->join[0], side->stream[0]) == hipSuccess &&
                      hipStreamWaitEvent(s, side->join[0], 0) == hipSuccess;
  if (!joined && rc == DX_OK) rc = fail(DX_EHIP, "forward: cannot join the side stream");
  return rc;
}

// observations (B,H,W,4) uint8 or float32 NHWC [optionally gathered by sample_idx] ->
// ctx->head (B,32): columns 0..A-1 logits, column A value.  Keeps y0,y1,y2,hid for backward.
int dx_cnn_forward(const dx_cnn_ctx *c, const void *obs, int obs_is_u8, const int32_t *sample_idx,
                   int B, void *stream) {
  DX_TRACE("dx_cnn_forward");
  if (int rc = check_ctx(c, "dx_cnn_forward", B, false)) return rc;
  DX_REQUIRE(obs != nullptr, "dx_cnn_forward: null observations");
  return forward_stages(c, ST_CONV0_FWD, ST_HEADS_FWD, obs, obs_is_u8, sample_idx, B, as_stream(stream));
}

// conv0 .. linear layer only (ctx->hid): the forward of an update whose heads run inside
// dx_cnn_heads_loss_f32
int dx_cnn_forward_trunk(const dx_cnn_ctx *c, const void *obs, int obs_is_u8, const int32_t *sample_idx,
                         int B, void *stream) {
  DX_TRACE("dx_cnn_forward_trunk");
  if (int rc = check_ctx(c, "dx_cnn_forward_trunk", B, false)) return rc;
  DX_REQUIRE(obs != nullptr, "dx_cnn_forward_trunk: null observations");
  // with the factored tail the trunk ends at y2: dx_cnn_heads_loss_f32 reads it directly
  return forward_stages(c, ST_CONV0_FWD, fc_factored(c) ? ST_CONV2_FWD : ST_FC_FWD, obs, obs_is_u8, sample_idx, B,
                        as_stream(stream));
}

// ctx->hid (B,512) -> ctx->head, the loss scalars, ctx->dhead, ctx->dhid and the heads' weight /
// bias gradient slabs, in ONE launch (heads.hip: heads_loss_fused_kernel).  DX_ENOSUP for more than
// 7 actions: the caller then uses dx_cnn_forward + dx_categorical_loss_f32 + dx_cnn_backward.
int dx_cnn_heads_loss_f32(const dx_cnn_ctx *c, const int64_t *actions, const float *old_log_prob,
                          const float *advantages, const float *old_values, const float *value_targets,
                          const double *norm_stats, float norm_eps, float *adv_normalized_out, int B, int mode,
                          float cliprange, float value_loss_coef, float entropy_coef, long long global_batch,
                          double *partials, int partials_capacity, unsigned *counter, float *loss_out,
                          void *stream) {
  DX_TRACE("dx_cnn_heads_loss_f32");
  if (int rc = check_ctx(c, "dx_cnn_heads_loss_f32", B, true)) return rc;
  if (fc_factored(c)) {  // out = y2 Wc^T + beff, the loss and dL/dout in one launch (heads.hip: tail_loss_kernel)
    g_route[ST_FC_FWD] = g_route[ST_HEADS_FWD] = g_route[ST_HEADS_WGRAD] = g_route[ST_HEADS_DGRAD] = "tail_factored";
    if (tail_fused(c)) {  // ... and dy2 + the partial G / s from the same pass over y2: backward_stages skips tail_bwd
      const Plan plan = make_plan(c, B);
      float *scratch = tail_scratch(c, plan, B);
      if (scratch == nullptr) return DX_EINVAL;
      const TailBwdPlan tb = tail_bwd_plan(scratch, B, c->num_actions);
      return launch_tail_loss_bwd(c->y2, c->packed + c->pk_wc, c->packed + c->pk_beff, actions, old_log_prob, advantages,
                                  old_values, value_targets, norm_stats, norm_eps, adv_normalized_out, c->head, c->dhead, B,
                                  c->num_actions, mode, cliprange, value_loss_coef, entropy_coef, global_batch, partials,
                                  partials_capacity, counter, loss_out, c->dy2, tb.gslab, tb.sslab, tb.Jp, tb.nwg,
                                  tb.rows_per_wg, as_stream(stream));
    }
    return launch_tail_loss(c->y2, c->packed + c->pk_wc, c->packed + c->pk_beff, actions, old_log_prob, advantages,
                            old_values, value_targets, norm_stats, norm_eps, adv_normalized_out, c->head, c->dhead, B,
                            c->num_actions, mode, cliprange, value_loss_coef, entropy_coef, global_batch, partials,
                            partials_capacity, counter, loss_out, as_stream(stream));
  }
  const Plan plan = make_plan(c, B);
  g_route[ST_HEADS_FWD] = g_route[ST_HEADS_WGRAD] = g_route[ST_HEADS_DGRAD] = "heads_loss_fused";
  return launch_heads_loss_fused(c->hid, c->packed + c->pk_hdf, c->packed + c->pk_hdb, actions, old_log_prob,
                                 advantages, old_values, value_targets, norm_stats, norm_eps, adv_normalized_out,
                                 c->head, c->dhead, c->dhid, c->slabs + plan.s[L_HD].w_off,
                                 c->slabs + plan.s[L_HD].b_off, plan.s[L_HD].msplit, plan.s[L_HD].mper, B,
                                 c->num_actions, mode, cliprange, value_loss_coef, entropy_coef, global_batch,
                                 partials, partials_capacity, counter, loss_out, as_stream(stream));
}

// DX_BWD_OVERLAP: the weight-gradient stages of the linear layer and of conv2 / conv1 on a side
// stream beside the chain of data gradients (each needs only its layer's output gradient, which
// the chain has produced by then; the finalisation waits for both): the two streams' kernels fill
// each other's ramps, last tile rounds and tails.  Measured per PPO iteration, same box,
// alternating: 256 envs (minibatch 8192) 45.4 -> 43.9 ms, 128 envs 26.3 -> 25.3 ms, 32 envs
// (minibatch 1024) 11.53 -> 11.61 ms -- there a stage is a single short round of tiles and the
// extra event traffic costs more than the overlap returns.  -1 (default) = for minibatches of at
// least DX_BWD_OVERLAP_MIN (2048) samples; 0 = never, 1 = always.  The kernels and their results
// are the same either way.  Round 4: with the image-resident bf16 kernels (wgrad_b6.hip, dgrad_b6.hip) every conv
// stage of the backward holds one workgroup per CU with most of its LDS -- two of them cannot share a CU, and
// the side stream returned nothing (23.8 vs 23.6-23.8 ms per iteration, alternating): the default then is
// the serial order.
static bool bwd_overlap(int B) {
  const int mode = DX_ENV("DX_BWD_OVERLAP", -1), limit = DX_ENV("DX_BWD_OVERLAP_MIN", 2048);
  return mode == 1 || (mode == -1 && B >= limit && !(wgrad_b6_on() && dgrad_b6_on()));
}

// Stages first .. last of the backward on `s`, the overlapped ones on the side stream, then the
// finalisation `which` (finalize_grads).  With the side stream and the whole gradient to finalise
// (which == 3) the slabs of everything but conv0 are reduced ON the side stream, under the first
// conv layer's weight gradient (the last stage of the chain), and only conv0's own slabs after it.
// slabs -> gradients for the bits of `which` (finalize_grads); with the factored tail bit 2 (linear layer +
// heads) is tail.hip's G reduction instead of a slab reduction
static int finalize_any(const dx_cnn_ctx *c, const Plan &plan, int which, int B, bool factored, hipStream_t s) {
  if (factored && (which & 2)) {
    float *scratch = tail_scratch(c, plan, B);
    if (scratch == nullptr) return DX_EINVAL;
    if ((which & ~2) != 0 && DX_ENV("DX_FINALIZE_MERGED", 1) != 0) {
      // the tail's G / s reduction rides in the conv layers' slab reduction (two independent slab readers, one
      // launch), the gradient products follow.  DX_FINALIZE_MERGED=0: three launches
      const TailGreduceArgs g = tail_greduce_args(scratch, B, c->num_actions);
      if (int rc = finalize_grads(c, plan, which & ~2, s, &g)) return rc;
      return launch_tail_grads(c->params, c->grads, c->off_w, c->off_b, c->num_actions, scratch, B, s, true);
    }
    if (int rc = launch_tail_grads(c->params, c->grads, c->off_w, c->off_b, c->num_actions, scratch, B, s)) return rc;
    which &= ~2;
  }
  return which != 0 ? finalize_grads(c, plan, which, s) : DX_OK;
}

static int backward_stages(const dx_cnn_ctx *c, int first, int last, const void *obs, int obs_is_u8,
                           const int32_t *sample_idx, int B, const Plan &plan, int which, hipStream_t s,
                           bool factored = false) {
  SideStream *side = bwd_overlap(B) ? side_stream() : nullptr;
  bool forked = false;
  int rc = DX_OK;
  for (int st = first; st <= last && rc == DX_OK; ++st) {
    if (factored && (st == ST_FC_WGRAD || st == ST_FC_DGRAD)) {
      // the factored tail: ONE pass over y2 gives dy2 and the partial G / s (tail.hip) in the place of the
      // linear layer's weight- and data-gradient GEMMs
      if (st == ST_FC_WGRAD) {
        g_route[ST_FC_WGRAD] = g_route[ST_FC_DGRAD] = "tail_factored";
        if (tail_fused(c)) continue;  // dx_cnn_heads_loss_f32 has formed dy2 and the partial G / s already
        float *scratch = tail_scratch(c, plan, B);
        rc = scratch == nullptr ? DX_EINVAL
                                : launch_tail_bwd(c->y2, c->dhead, c->packed + c->pk_wc, c->dy2, scratch, B, c->num_actions, s);
      }
      continue;
    }
    // DX_BWD_OVERLAP_MASK: which weight-gradient stages go aside (1 = linear layer, 2 = conv2, 4 = conv1)
    const int aside_mask = DX_ENV("DX_BWD_OVERLAP_MASK", 7);
    const bool aside = side != nullptr && ((st == ST_FC_WGRAD && (aside_mask & 1)) || (st == ST_CONV2_WGRAD && (aside_mask & 2)) ||
                                            (st == ST_CONV1_WGRAD && (aside_mask & 4)));
    if (aside) {  // everything enqueued on `s` so far (this layer's output gradient) comes first
      if (hipEventRecord(side->fork, s) != hipSuccess || hipStreamWaitEvent(side->stream[0], side->fork, 0) != hipSuccess) {
        rc = fail(DX_EHIP, "backward: cannot order the side stream");
        break;
      }
      forked = true;
    }
    // the heads' slabs (heads stages or dx_cnn_heads_loss_f32, on `s`) are ordered before the side
    // stream's finalisation by the fork of the linear layer's weight gradient
    rc = run_stage(c, st, obs, obs_is_u8, sample_idx, B, plan, aside ? side->stream[0] : s);
    const bool side_finalize = DX_ENV("DX_BWD_SIDE_FINALIZE", 1) != 0;
    if (rc == DX_OK && aside && side_finalize && st == ST_CONV1_WGRAD && which == 3 && last == ST_CONV0_WGRAD) {
      rc = finalize_any(c, plan, 2 | 8, B, factored, side->stream[0]);
      which = 4;
    }
  }
  if (forked) {  // also after a failed launch: the caller's stream never runs ahead of the side stream
    const bool joined = hipEventRecord(side->join[0], side->stream[0]) == hipSuccess &&
                        hipStreamWaitEvent(s, side->join[0], 0) == hipSuccess;
    if (!joined && rc == DX_OK) rc = fail(DX_EHIP, "backward: cannot join the side stream");
  }
  if (rc == DX_OK && which != 0) rc = finalize_any(c, plan, which, B, factored, s);
  return rc;
}

// ctx->dhead (B,32) -> ctx->grads (canonical layout), using the activations kept by
// dx_cnn_forward on the SAME observations / sample_idx.
int dx_cnn_backward(const dx_cnn_ctx *c, const void *obs, int obs_is_u8, const int32_t *sample_idx,
                    int B, void *stream) {
  DX_TRACE("dx_cnn_backward");
  if (int rc = check_ctx(c, "dx_cnn_backward", B, true)) return rc;
  DX_REQUIRE(obs != nullptr, "dx_cnn_backward: null observations");
  const Plan plan = make_plan(c, B);
  g_route[ST_FINALIZE] = "finalize";
  return backward_stages(c, ST_HEADS_WGRAD, ST_CONV0_WGRAD, obs, obs_is_u8, sample_idx, B, plan, 3, as_stream(stream));
}

// The same backward in two calls so that a data-parallel caller can start the all-reduce of the
// big tail of the gradient buffer early: part 0 = heads + linear layer (their gradients
// grads[off_w[3] .. param_count) are final when it returns to the stream), part 1 = the three
// conv layers (grads[0 .. off_w[3])).  Part 0 must be enqueued first.  After dx_cnn_heads_loss_f32
// (which has done the heads' share): part 2 = the linear layer only in part 0's place, part 3 =
// linear layer + conv layers with one finalisation (the single-process flow).
int dx_cnn_backward_part(const dx_cnn_ctx *c, const void *obs, int obs_is_u8, const int32_t *sample_idx,
                         int B, int part, void *stream) {
  DX_TRACE("dx_cnn_backward_part");
  if (int rc = check_ctx(c, "dx_cnn_backward_part", B, true)) return rc;
  DX_REQUIRE(obs != nullptr, "dx_cnn_backward_part: null observations");
  DX_REQUIRE(part >= 0 && part <= 3, "dx_cnn_backward_part: part must be 0..3, got %d", part);
  const Plan plan = make_plan(c, B);
  hipStream_t s = as_stream(stream);
  // parts 2 / 3: the heads' dgrad and weight-gradient slabs already exist (dx_cnn_heads_loss_f32)
  const int first = part == 0 ? ST_HEADS_WGRAD : part == 1 ? ST_CONV2_WGRAD : ST_FC_WGRAD;
  const int last = (part == 0 || part == 2) ? ST_FC_DGRAD : ST_CONV0_WGRAD;
  // parts 2 / 3 follow dx_cnn_heads_loss_f32: with the factored tail that call left dL/dout in ctx->dhead and
  // the linear layer's share of the backward is tail.hip's
  const bool factored = (part == 2 || part == 3) && fc_factored(c);
  return backward_stages(c, first, last, obs, obs_is_u8, sample_idx, B, plan, part == 1 ? 1 : part == 3 ? 3 : 2, s,
                         factored);
}

// DX_FC_ROLLOUT=0: the rollout's linear layer on the 32x32-tile split-K latency kernel (igemm_lat.hip)
// instead of the weight-stationary kernel of fc_rollout.hip; DX_FC_ROLLOUT_MAX_B: largest batch it takes
static bool fc_rollout_on(int B, int flat) {
  const int on = DX_ENV("DX_FC_ROLLOUT", 1), max_b = DX_ENV("DX_FC_ROLLOUT_MAX_B", 1024);
  return on != 0 && B <= max_b && fc_rollout_supported(B, kHid, flat);
}

// split of the 3136-deep linear layer over K for small batches (98 K-steps = 2 x 7 x 7)
static int fc_ksplit(int B, int flat) {
  if (fc_rollout_on(B, flat)) return fc_rollout_parts();
  const int steps = flat / 64 * 64 == flat ? flat / 64 : flat / 32;  // 64-deep K steps when possible
  int ks = B <= 1024 ? 7 : 1;
  while (ks > 1 && steps % ks) --ks;
  return ks;
}

// DX_CONVSTACK=0: the rollout's conv layers as three launches (igemm_lat.hip / conv0_b16.hip) instead of
// the one-workgroup-per-image launch of convstack.hip; DX_CONVSTACK_MAX_B: largest batch it takes
static bool convstack_on(int B) {
  const int on = DX_ENV("DX_CONVSTACK", 1), max_b = DX_ENV("DX_CONVSTACK_MAX_B", 1024);
  return on != 0 && B <= max_b;
}

// the rollout takes the factored tail wherever the training path does
static bool act_factored(const dx_cnn_ctx *c) { return fc_factored(c); }

// the conv-stack kernel's operands from the ctx (convstack.hip)
static ConvStackArgs convstack_args(const dx_cnn_ctx *c, uint8_t *obs, int B) {
  ConvStackArgs a;
  std::memset(&a, 0, sizeof(a));
  a.obs = obs;
  a.Wb0 = planes(c, c->pb_c0f); a.bias0 = c->params + c->off_b[0];
  a.Wf1 = planes(c, c->ps_c1f); a.bias1 = c->params + c->off_b[1];
  a.Wf2 = planes(c, c->ps_c2f); a.bias2 = c->params + c->off_b[2];
  a.B = B; a.T = 1; a.row_stride = B;
  return a;
}
static bool convstack_usable(const dx_cnn_ctx *c, int obs_is_u8, int B) {
  return obs_is_u8 && !conv0_f32() && convstack_supported(c->in_h, c->in_w, c->in_c) && convstack_on(B);
}
// the tail of the policy inside the conv-stack launch: out = y2 Wc^T + beff and the sampling rule
static void convstack_tail(const dx_cnn_ctx *c, ConvStackArgs *a, const float *uniforms, uint64_t seed, uint64_t counter,
                           int env0, int64_t *actions, float *log_prob, float *values) {
  a->Wc = c->packed + c->ps_wc; a->beff = c->packed + c->pk_beff; a->A = c->num_actions;
  a->uniforms = uniforms; a->seed = seed; a->counter = counter; a->env0 = env0;
  a->actions = actions; a->log_prob = log_prob; a->values = values;
}

// conv0 .. linear layer (split-K slabs) of a rollout step; the caller finishes with a heads launch
static int act_trunk(const dx_cnn_ctx *c, const void *obs, int obs_is_u8, int B, NTArgs *fc, hipStream_t s) {
  const Plan plan = make_plan(c, B);
  if (convstack_usable(c, obs_is_u8, B)) {
    // the three conv layers of every image in ONE launch, y0 / y1 never leave the CU (convstack.hip)
    ConvStackArgs cs = convstack_args(c, static_cast<uint8_t *>(const_cast<void *>(obs)), B);
    cs.y2 = c->y2;
    if (int rc = launch_convstack(cs, s)) return rc;
    g_route[ST_CONV0_FWD] = g_route[ST_CONV1_FWD] = g_route[ST_CONV2_FWD] = "convstack";
  } else {
    for (int st = ST_CONV0_FWD; st <= ST_CONV2_FWD; ++st)
      if (int rc = run_stage(c, st, obs, obs_is_u8, nullptr, B, plan, s)) return rc;
  }
  if (act_factored(c)) return DX_OK;  // the caller finishes with tail_act: out = y2 Wc^T + beff (no 512-wide layer at all)
  const int ks = fc_ksplit(B, c->flat);
  DX_REQUIRE(static_cast<long long>(ks) * B * kHid <= c->hid_slab_count, "dx_cnn_act: hid_slabs too small");
  *fc = nt_args(rows_gather(c->y2, c->flat), c->packed + c->pk_fcf, c->params + c->off_b[3],
                c->hid_slabs, kHid, B, kHid, c->flat);
  fc->ksplit = ks;
  fc->slab_stride = static_cast<long long>(B) * kHid;
  if (fc_rollout_on(B, c->flat)) {  // every weight read once per launch (fc_rollout.hip)
    g_route[ST_FC_FWD] = "fc_rollout";
    return launch_fc_rollout(c->y2, c->packed + c->pk_fcf, c->params + c->off_b[3], c->hid_slabs, B, s);
  }
  g_route[ST_FC_FWD] = "igemm_nt split-K";
  return launch_nt(*fc, false, EPI_BIAS, ST_FC_FWD, s);
}

int dx_cnn_act(const dx_cnn_ctx *c, const void *obs, int obs_is_u8, int B, const float *uniforms,
               uint64_t seed, uint64_t counter, int64_t *actions, float *log_prob, float *values,
               void *stream) {
  DX_TRACE("dx_cnn_act");
  if (int rc = check_ctx(c, "dx_cnn_act", B, false)) return rc;
  DX_REQUIRE(obs != nullptr && c->hid_slabs != nullptr, "dx_cnn_act: null observations / hid_slabs");
  hipStream_t s = as_stream(stream);
  if (convstack_usable(c, obs_is_u8, B) && act_factored(c)) {
    // the WHOLE act step in one launch: conv stack, y2 Wc^T + beff and the sampling, one workgroup per image
    ConvStackArgs cs = convstack_args(c, static_cast<uint8_t *>(const_cast<void *>(obs)), B);
    cs.y2 = c->y2;  // (kept: tests and callers may look at the conv output of the last act)
    convstack_tail(c, &cs, uniforms, seed, counter, 0, actions, log_prob, values);
    for (int st = ST_CONV0_FWD; st <= ST_CONV2_FWD; ++st) g_route[st] = "convstack";
    g_route[ST_FC_FWD] = g_route[ST_HEADS_FWD] = "convstack (tail_factored)";
    return launch_convstack(cs, s);
  }
  NTArgs a;
  if (int rc = act_trunk(c, obs, obs_is_u8, B, &a, s)) return rc;
  if (act_factored(c)) {
    g_route[ST_FC_FWD] = g_route[ST_HEADS_FWD] = "tail_factored";
    return launch_tail_act(c->y2, c->packed + c->pk_wc, c->packed + c->pk_beff, B, c->num_actions, uniforms, seed,
                           counter, actions, log_prob, values, s);
  }
  return launch_heads_act_fused(c->hid_slabs, a.ksplit, a.slab_stride, c->packed + c->pk_hdf,
                                c->packed + c->pk_hdb, B, c->num_actions, uniforms, seed, counter,
                                actions, log_prob, values, s);
}

// T rollout steps against the synthetic device env enqueued from one call.  Per step: the policy
// trunk (4 launches) and ONE launch that finishes the policy (slab sum + heads + sampling) and
// generates the next observation batch, rewards and resets -- bit-identical to dx_cnn_act +
// dx_synth_atari_step (the synthetic dynamics ignore the action, so the env's part of the grid
// does not wait for the policy's), one dependent launch shorter.  Buffers are time-major:
// obs (T+1, N, H, W, 4) uint8 with obs[0] given, actions (T, N) int64, log_prob / values /
// rewards (T, N) float32, resets (T, N) bytes.
int dx_cnn_rollout_synth(const dx_cnn_ctx *c, uint8_t *obs, int T, int N, int64_t *actions,
                         float *log_prob, float *values, float *rewards, uint8_t *resets,
                         uint64_t policy_seed, uint64_t policy_counter, uint64_t env_seed,
                         uint64_t env_counter, float p_reward, float p_reset, void *stream) {
  DX_TRACE("dx_cnn_rollout_synth");
  if (int rc = check_ctx(c, "dx_cnn_rollout_synth", N, false)) return rc;
  DX_REQUIRE(T >= 1 && obs && actions && log_prob && values && rewards && resets && c->hid_slabs,
             "dx_cnn_rollout_synth: bad arguments");
  const long long frame = static_cast<long long>(c->in_h) * c->in_w * c->in_c * N;
  DX_REQUIRE(frame % 16 == 0, "dx_cnn_rollout_synth: frame batch must be a multiple of 16 bytes");
  hipStream_t s = as_stream(stream);
  if (convstack_usable(c, 1, N) && act_factored(c)) {
    // ONE launch for the whole horizon: every env's chain frame -> conv stack -> y2 Wc^T -> sample -> next frame
    // is local to one workgroup (the synthetic env is a hash of (seed, counter, position)), so nothing crosses
    // workgroups between steps: conv0's weights stay in LDS, the next layer's weight fragments and the next
    // frame travel / are generated under the matrix instructions, and 129 x 2-5 kernel boundaries are gone.
    // The same kernel with T = 1 is dx_cnn_act: the buffers equal the per-step loop's bit for bit.
    ConvStackArgs cs = convstack_args(c, obs, N);
    convstack_tail(c, &cs, nullptr, policy_seed, policy_counter, 0, actions, log_prob, values);
    cs.env = 1; cs.T = T; cs.row_stride = N;
    cs.rewards = rewards; cs.resets = resets; cs.env_seed = env_seed; cs.env_counter = env_counter;
    cs.p_reward = p_reward; cs.p_reset = p_reset;
    for (int st = ST_CONV0_FWD; st <= ST_CONV2_FWD; ++st) g_route[st] = "convstack";
    g_route[ST_FC_FWD] = g_route[ST_HEADS_FWD] = "convstack (tail_factored)";
    return launch_convstack(cs, s);
  }
  // Lanes: the envs are cut into equal parts whose steps are independent chains (part A's step
  // t + 1 needs only part A's step t), enqueued on the caller's stream and on side streams, so
  // that one part's launch tails and ramps overlap the others' compute.  Frames, rewards and resets
  // are bit-identical to the one-lane rollout and the samples come from the same stream positions
  // (both are indexed by the env's position in the whole batch); log-probs and values agree to
  // float32 rounding (a half batch may take a different tile shape).  DX_ROLLOUT_LANES=1: off.
  // Lanes need >= 64 envs each.  Smaller lanes were measured (round 3, tools/lanes_probe.sh): a
  // 32-env shard as 2 / 4 chains of 16 / 8 envs takes 12.7 / 14.3 ms per iteration against 11.8 as
  // one chain -- kernels of different queues do not fill each other's dependent-launch gaps at
  // this size, and the host pays 5.5 / 15.7 ms of launches instead of 3.5.
  int lanes = rollout_lanes() < kMaxLanes ? rollout_lanes() : kMaxLanes;
  while (lanes > 1 && !(N / lanes >= rollout_lane_min() && N % (4 * lanes) == 0 && (frame / lanes) % 16 == 0 &&
                        (act_factored(c) ||
                         static_cast<long long>(lanes) * fc_ksplit(N / lanes, c->flat) * (N / lanes) * kHid <= c->hid_slab_count)))
    --lanes;
  SideStream *side = lanes > 1 ? side_stream() : nullptr;
  if (lanes > 1 && side == nullptr) return DX_EHIP;
  const int part = N / lanes;
  const long long pframe = frame / lanes;
  dx_cnn_ctx lane_ctx[kMaxLanes];
  for (int l = 0; l < lanes; ++l) {
    dx_cnn_ctx &b = lane_ctx[l];
    b = *c;
    const long long first = static_cast<long long>(l) * part;
    b.y0 += first * c->h0 * c->w0 * kC0;
    b.y1 += first * c->h1 * c->w1 * kC1;
    b.y2 += first * c->flat;
    b.hid_slabs += fc_ksplit(part, c->flat) * first * kHid;
    b.hid_slab_count -= fc_ksplit(part, c->flat) * first * kHid;
  }
  if (lanes > 1) {
    DX_HIP(hipEventRecord(side->fork, s));
    for (int l = 1; l < lanes; ++l) DX_HIP(hipStreamWaitEvent(side->stream[l - 1], side->fork, 0));
  }
  int rc = DX_OK;
  for (int t = 0; t < T && rc == DX_OK; ++t) {
    for (int l = 0; l < lanes && rc == DX_OK; ++l) {
      hipStream_t ls = l == 0 ? s : side->stream[l - 1];
      const dx_cnn_ctx *lc = &lane_ctx[l];
      NTArgs a;
      rc = act_trunk(lc, obs + t * frame + l * pframe, 1, part, &a, ls);
      if (rc != DX_OK) break;
      const long long row = static_cast<long long>(t) * N + l * part;
      if (act_factored(c)) {
        rc = launch_tail_act_synth(lc->y2, c->packed + c->pk_wc, c->packed + c->pk_beff, part, c->num_actions,
                                   policy_seed, policy_counter + t, actions + row, log_prob + row, values + row,
                                   obs + (t + 1) * frame + l * pframe, pframe, rewards + row, resets + row, env_seed,
                                   env_counter + t, p_reward, p_reset, l * part, l * (pframe / 16), ls);
        continue;
      }
      rc = launch_heads_act_synth(lc->hid_slabs, a.ksplit, a.slab_stride, c->packed + c->pk_hdf,
                                  c->packed + c->pk_hdb, part, c->num_actions, policy_seed,
                                  policy_counter + t, actions + row, log_prob + row, values + row,
                                  obs + (t + 1) * frame + l * pframe, pframe, rewards + row,
                                  resets + row, env_seed, env_counter + t, p_reward, p_reset, l * part,
                                  l * (pframe / 16), ls);
    }
  }
  // also on failure: the caller's stream must not run ahead of (and the caller must not free
  // buffers under) work already enqueued on the side streams
  for (int l = 1; l < lanes; ++l) {
    const bool joined = hipEventRecord(side->join[l - 1], side->stream[l - 1]) == hipSuccess &&
                        hipStreamWaitEvent(s, side->join[l - 1], 0) == hipSuccess;
    if (!joined) {
      (void)hipStreamSynchronize(side->stream[l - 1]);
      if (rc == DX_OK) rc = fail(DX_EHIP, "dx_cnn_rollout_synth: cannot join the side streams");
    }
  }
  return rc;
}

// 1 when this ctx's updates and rollouts run the linear layer + heads as one affine map of y2 (tail.hip)
int dx_cnn_tail_factored(const dx_cnn_ctx *c) {
  return (c != nullptr && c->struct_bytes == static_cast<int>(sizeof(dx_cnn_ctx)) && fc_factored(c)) ? 1 : 0;
}

// 1 when dx_cnn_heads_loss_f32 applies to this ctx (the heads, the loss and the heads' backward in one launch): up to
// 18 actions where the tail is factored (tail.hip), up to 7 on the layer-by-layer route (the head rows in registers)
int dx_cnn_fused_heads(const dx_cnn_ctx *c) {
  if (c == nullptr || c->struct_bytes != static_cast<int>(sizeof(dx_cnn_ctx))) return 0;
  return (fc_factored(c) || c->num_actions + 1 <= 8) ? 1 : 0;
}

// 1: dx_cnn_heads_loss_f32 also runs the factored tail's backward pass over y2 (tail_fused above)
int dx_cnn_tail_fused(const dx_cnn_ctx *c) {
  if (c == nullptr || c->struct_bytes != static_cast<int>(sizeof(dx_cnn_ctx))) return 0;
  return tail_fused(c) ? 1 : 0;
}

// Kernel family the LAST launch of `stage` took ("ntp", "wgrad_direct", "igemm_pix", ...; "" before any).
const char *dx_cnn_last_route(int stage) {
  return (stage >= 0 && stage < ST_COUNT && g_route[stage]) ? g_route[stage] : "";
}

// A single stage, for per-kernel timing (bench.py roofline) and layer-level tests.
int dx_cnn_stage(const dx_cnn_ctx *c, int stage, const void *obs, int obs_is_u8,
                 const int32_t *sample_idx, int B, void *stream) {
  DX_TRACE("dx_cnn_stage");
  if (int rc = check_ctx(c, "dx_cnn_stage", B, stage >= ST_HEADS_WGRAD)) return rc;
  DX_REQUIRE(obs != nullptr, "dx_cnn_stage: null observations");
  const Plan plan = make_plan(c, B);
  return run_stage(c, stage, obs, obs_is_u8, sample_idx, B, plan, as_stream(stream));
}

}  // extern "C"

// ---- for dx_cnn_ppo_epoch (cnn_epoch.hip): pieces of the calls above, and the side stream ----
namespace dx {

// forward stages first .. last (ST_CONV0_FWD .. ST_HEADS_FWD) of a minibatch already checked by the caller
int cnn_forward_range(const dx_cnn_ctx *c, int first, int last, const void *obs, int obs_is_u8,
                      const int32_t *sample_idx, int B, hipStream_t s) {
  return forward_stages(c, first, last, obs, obs_is_u8, sample_idx, B, s);
}

int cnn_pack_part(const dx_cnn_ctx *c, int part, hipStream_t s) { return pack_part(c, part, s); }

// the mirrors between two updates of a native epoch: with the factored tail everything but the linear
// layer's GEMM mirrors (nothing reads them until the epoch's last update has packed them again)
int cnn_pack_between_updates(const dx_cnn_ctx *c, bool last_update, int obs_is_u8, hipStream_t s) {
  const bool light = fc_factored(c) && !last_update;
  // (the direct pack: only when the NEXT minibatch's every stage reads the bf16 planes -- uint8 frames through the
  // one-launch forward, bf16 data gradients, the first layer's bf16 kernels; DX_PACK_DIRECT=0: the general packs)
  const bool direct_env = DX_ENV("DX_PACK_DIRECT", 1) != 0;
  const bool direct = light && direct_env && convstack_train_usable(c, obs_is_u8) && dgrad_b6_usable(c) && wgrad_b6_on() &&
                      c->in_c == 4 && conv0_direct_supported(c->in_h, c->in_w, c->in_c, c->h0, c->w0);
  return pack_part(c, 3, s, light, direct);
}
bool cnn_fc_factored(const dx_cnn_ctx *c) { return fc_factored(c); }
// the whole conv stack of a training minibatch is one launch: no first layer to start ahead of the other mirrors
bool cnn_forward_fused(const dx_cnn_ctx *c, int obs_is_u8) { return convstack_train_usable(c, obs_is_u8); }

// the library's side stream, ordered after everything enqueued on `s` so far (NULL: unavailable)
hipStream_t cnn_side_begin(hipStream_t s) {
  SideStream *side = side_stream();
  if (side == nullptr) return nullptr;
  if (hipEventRecord(side->fork, s) != hipSuccess || hipStreamWaitEvent(side->stream[0], side->fork, 0) != hipSuccess) {
    fail(DX_EHIP, "cnn_side_begin: cannot order the side stream");
    return nullptr;
  }
  return side->stream[0];
}

// `s` ordered after everything enqueued on the side stream
int cnn_side_end(hipStream_t s) {
  SideStream *side = side_stream();
  if (side == nullptr) return DX_EHIP;
  if (hipEventRecord(side->join[0], side->stream[0]) != hipSuccess || hipStreamWaitEvent(s, side->join[0], 0) != hipSuccess) {
    (void)hipStreamSynchronize(side->stream[0]);
    return fail(DX_EHIP, "cnn_side_end: cannot join the side stream");
  }
  return DX_OK;
}

}  // namespace dx


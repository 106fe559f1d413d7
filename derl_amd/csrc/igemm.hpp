// Implicit-GEMM building blocks for the Nature-DQN conv stack on fp32 MFMA (gfx950).
//
// Activations are pixel-major NHWC ([img][y][x][c], fp32; the observation may be uint8),
// so the K axis of every contraction is cut into `nseg` contiguous runs of `seglen`
// elements: a forward conv row is KH runs of KW*IC elements, a dgrad row is one run of OC
// elements per kernel tap, a linear layer is a single run.  Weights are read from packed
// mirrors laid out [N][K] in that same K order (dx_permute_reduce builds them from the
// canonical state_dict layout).
#pragma once
#include "common.hpp"

namespace dx {

// exact n / d for n < 2^31 (Granlund-Montgomery, N = 31): q = umulhi(n, magic) >> shift
struct FastDiv {
  uint32_t magic, shift, d;
};

inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{0u, 0u, d};
  if (d <= 1) return f;
  uint32_t L = 0;
  while ((1ull << L) < d) ++L;
  f.magic = static_cast<uint32_t>((1ull << (31 + L)) / d + 1);
  f.shift = L - 1;
  return f;
}

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
  const uint32_t q = __umulhi(n, f.magic) >> f.shift;  // computed unconditionally: no branch
  return f.d <= 1 ? n : q;
}

constexpr int kMaxSeg = 16;

// Row m of the implicit A matrix: m -> (img, oy, ox); run s starts at source pixel
// (oy*sy + dy[s], ox*sx + dx[s]), channel coff[s], and is `seglen` contiguous elements.
struct Gather {
  const void *src;       // NHWC activations (float or uint8)
  const int32_t *idx;    // optional image gather (minibatch by permutation), or nullptr
  long long img_stride;  // H*W*C elements
  int H, W, C;
  FastDiv div_img, div_row;  // by OH*OW and by OW
  int OHW, OW;
  int sy, sx;
  int nseg, seglen;
  int check;             // runs may start outside the image -> zero fill
  int seg_off[kMaxSeg];  // (dy*W + dx)*C + coff
  int seg_dy[kMaxSeg], seg_dx[kMaxSeg];  // dwords: read with scalar loads
};

// Output address of GEMM element (m, n); identity (m*ldc + n) unless `enabled`.  Enabled:
// row m -> (img, oy, ox) and column n -> (group g = n / chan, channel n % chan) with
// g -> (py, px) = (g / osx, g % osx); the element goes to pixel (oy*osy + py, ox*osx + px) of
// an (OUT_H, OUT_W, ldc) image if that pixel exists.  This is the strided-conv dgrad: all
// osy*osx parity classes of the input pixels share one gathered A matrix.
struct OutMap {
  int enabled;
  FastDiv div_img, div_row;
  int OHW, OW;
  int OUT_H, OUT_W, osy, osx, chan;
};

// EPI_MASK: relu' from the kept output; EPI_BIAS_TANH / EPI_DTANH: tanh and tanh' = 1 - y^2
enum Epilogue { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_MASK = 3, EPI_BIAS_TANH = 4, EPI_DTANH = 5 };

// C[m][n] = epi(sum_k A(m,k) * Wp[n][k])
struct NTArgs {
  Gather g;
  OutMap om;
  const float *Wp;        // [N][K]
  const uint16_t *Wb;     // optional: the same matrix as three bf16 planes [3][N][K] (hi, mid, lo)
  long long wb_plane;     // elements per plane (N*K)
  const float *bias;      // [N]
  const float *mask_src;  // EPI_MASK: zero where mask_src[out_row][n] <= 0
  float *out;
  int ldc;
  int M, N, K;
  int ksplit;             // > 1: raw partial sums into slabs out + z*slab_stride
  long long slab_stride;
};

// slab[z][n][k] = sum_{m in slice z} G[m][n] * A(m,k);  bias_slab[z][n] = sum_m G[m][n]
struct TNArgs {
  Gather g;
  const float *G;  // [M][ldg]
  int ldg;
  float *slab;     // [msplit][N][K]
  float *bias_slab;  // [msplit][N] or nullptr
  int M, N, K;
  int msplit, mper;  // mper % 32 == 0
  int swizzle, gk, gn;  // set by the launcher: XCD-aware 1-D block numbering (see igemm_tn_kernel)
};

// network stages: one kernel instantiation (= one profiler row) each
enum Stage {
  ST_CONV0_FWD = 0, ST_CONV1_FWD, ST_CONV2_FWD, ST_FC_FWD, ST_HEADS_FWD,
  ST_HEADS_WGRAD, ST_HEADS_DGRAD, ST_FC_WGRAD, ST_FC_DGRAD, ST_CONV2_WGRAD, ST_CONV2_DGRAD,
  ST_CONV1_WGRAD, ST_CONV1_DGRAD, ST_CONV0_WGRAD, ST_FINALIZE,
  // two-net tanh MLP (derl/models.py:224-271)
  ST_MLP_HIDDEN, ST_MLP_OUT, ST_MLP_DGRAD, ST_MLP_WGRAD_HID, ST_MLP_WGRAD_OUT, ST_COUNT
};

int launch_nt(const NTArgs &a, bool a_u8, int epi, int stage, hipStream_t stream);
int launch_tn(const TNArgs &a, bool a_u8, int stage, hipStream_t stream);
// dgrad GEMMs tiled as one pixel x 64 images so that zero taps are skipped (igemm_pix.hip)
int launch_nt_pix(const NTArgs &a, int nimg, int epi, int stage, hipStream_t stream);
// latency-shaped NT kernels for small batches (igemm_lat.hip); DX_ENOSUP = not covered
int launch_nt_lat(const NTArgs &a, bool a_u8, int epi, int stage, hipStream_t stream);
// 3xbf16-split variant of the big NT stages (igemm_b3.hip); DX_ENOSUP = not covered
int launch_nt_b3(const NTArgs &a, int epi, int stage, hipStream_t stream);
// persistent LDS-DMA ring kernels for the 64-column conv stages (ntp.hip); DX_ENOSUP = not covered
int launch_ntp_fwd(const NTArgs &a, int stage, hipStream_t stream);
int launch_ntp_pix(const NTArgs &a, int nimg, int TA, int TB, hipStream_t stream);
// ksplit_slabs (optional, ksplit_capacity floats): scratch for the forward's two K halves at small tile counts
int launch_ntp_rows(const float *A, int lda, const float *W, const float *mask, const float *bias, float *out, int M,
                    int N, int K, float *ksplit_slabs, long long ksplit_capacity, hipStream_t stream);

// plain row-major NT GEMM fed by LDS-DMA (nt_dma.hip): the linear layer's forward and dgrad
struct NtDmaArgs {
  const float *A;  // [M][lda]
  const float *W;  // [N][K]
  const float *bias;      // EPI_BIAS
  const float *mask_src;  // EPI_MASK, [M][ldc]
  float *out;      // [M][ldc]
  int M, N, K, lda, ldc;
};
bool nt_dma_supported(int M, int N, int K);
int launch_nt_dma(const NtDmaArgs &a, int epi, hipStream_t stream);
constexpr int kMaxSplitJobs = 8;
int launch_split_planes(const float *const *src, uint16_t *const *dst, const long long *count, int njobs,
                        hipStream_t stream);

// gather (scatter = 0): dst[i] = sum_z src[z*slab_stride + off + d0*s0 + d1*s1 + d2*s2 + d3*s3]
// scatter (scatter = 1): dst[d0*s0 + d1*s1 + d2*s2 + d3*s3] = sum_z src[z*slab_stride + off + i]
// with i = ((d0*D1+d1)*D2+d2)*D3+d3 the CONTIGUOUS side (slab reads stay coalesced when scattering)
struct PermuteJob {
  const float *src;
  float *dst;
  long long total;
  int D1, D2, D3;
  long long s0, s1, s2, s3, off;
  int nslab;
  long long slab_stride;
  int scatter;
};
constexpr int kMaxJobs = 24;
int launch_permute_reduce(const PermuteJob *jobs, int njobs, hipStream_t stream);
// the linear layer's two mirrors with coalesced reads and writes (row permute, then tiled transpose)
int launch_finalize_fused(const PermuteJob *jobs, int njobs, const float *slab, int nslab, long long slab_stride,
                          float *grad, int N, int P, int C, hipStream_t stream);
int launch_pack_fused(const PermuteJob *jobs, int njobs, const float *W, float *fcf, float *fcd, int N, int P, int C,
                      hipStream_t stream);
int launch_fc_pack(const float *W, float *fcf, float *fcd, int N, int P, int C, hipStream_t stream);
// canonical dW[n][c*P + p] = sum over slabs of slab[z][n][p*C + c] (coalesced on both sides)
int launch_fc_grad_finalize(const float *slab, int nslab, long long slab_stride, float *grad, int N, int P,
                            int C, hipStream_t stream);

// image-resident weight gradients of the 4x4/2 and 3x3/1 convolutions (wgrad_direct.hip): one
// persistent workgroup per slab, x = layer input [B][IH][IW][IC], g = output gradient [B][OH][OW][OC]
struct WgradDirectArgs {
  const float *x;
  const float *g;
  float *slab;       // [workgroups][OC][KH*KW*IC]
  float *bias_slab;  // [workgroups][OC]
  int B, IH, IW, OH, OW;
  int diag;  // timing experiments only (DX_WD_DIAG): bit 0 copy only the first image, bit 1 skip the slab store
  int descending;  // wgrad_b6.hip: walk the minibatch from its LAST image down (bwd_descending)
};
// Which way a backward stage walks the minibatch.  A stage's operands were written (or last read) by the stage before
// it, ascending; what the 256 MB last-level cache still holds of them is their TAIL.  So the weight gradients walk
// descending (conv1: the tail of dY1 that conv2's data gradient just wrote; conv0: the tail of dY0) and the data
// gradient that follows walks ascending again -- into the low images the weight gradient read last.  DX_BWD_ORDER:
// bit 0 conv2 wgrad, 1 conv2 dgrad, 2 conv1 wgrad, 3 conv1 dgrad, 4 conv0 wgrad descending (default 21 = 1 + 4 + 16).
bool bwd_descending(int stage);
bool wgrad_direct_supported(int IH, int IW, int IC, int OH, int OW, int OC, int KH, int KW, int S);
int launch_wgrad_direct(const WgradDirectArgs &a, int stage, int nwg, hipStream_t stream);
int wgrad_direct_workgroups(int stage, long long batch);  // persistent workgroups that fill the chip for this layer
// the same two weight gradients on the bf16 matrix cores, six exact products per fp32 x fp32 (wgrad_b6.hip)
// conv1 / conv2 data gradients the same way, image-resident (dgrad_b6.hip)
bool dgrad_b6_on();
long long dgrad_b6_pack_elems(int layer);
int launch_dgrad_b6_pack(const float *const c1d[4], const float *c2d, uint16_t *Wf1, uint16_t *Wf2, hipStream_t stream);
int launch_dgrad_b6(int layer, const float *g, const uint16_t *Wf, const float *mask_src, float *out, int B, hipStream_t stream,
                    bool descending = false);
bool wgrad_b6_on();
int launch_wgrad_b6(const WgradDirectArgs &a, int stage, int nwg, hipStream_t stream);

// weight gradient of the 512-wide linear layer (wgrad_fc.hip): slab[slice][512][K] partial sums
// over `msplit` row slices; the bias gradient partials come from launch_colsum
struct FcWgradArgs {
  const float *G;  // [M][512]
  const float *A;  // [M][K]
  float *slab;
  int M, K, msplit;
  int gk;          // set by the launcher
  int diag;        // timing experiments (DX_FC_DIAG): bit 0 no copies after the prologue, bit 1 no barrier, bit 2 no store
};
bool fc_wgrad_supported(int M, int N, int K);
int fc_wgrad_slices(int M, int K);
int launch_fc_wgrad(const FcWgradArgs &a, hipStream_t stream);
// out[chunk][N] = column sums of G over `chunks` row chunks
int launch_colsum(const float *G, float *out, int M, int N, int chunks, hipStream_t stream);

// direct first-layer convolution on uint8 frames (conv0.hip)
struct Conv0Args {
  const uint8_t *obs;     // (imgs, in_h, in_w, 4) uint8
  const int32_t *idx;     // optional image gather
  int in_h, in_w, h0, w0;
  int M, ntiles;          // output pixels = B*h0*w0, tiles of 256 pixels
  const float *Wp, *bias; // forward: packed [32][256], bias [32]
  float *out;             // forward: y0 [M][32]
  const float *G;         // wgrad: dY0 [M][32]
  float *slab, *bias_slab;  // wgrad: [nblocks][32][256], [nblocks][32]
};
bool conv0_direct_supported(int in_h, int in_w, int in_c, int h0, int w0);
int launch_conv0_fwd(const Conv0Args &a, hipStream_t stream);
int launch_conv0_wgrad(const Conv0Args &a, int nblocks, hipStream_t stream);
// the same two kernels on the bf16 matrix cores with an exact 3-term weight / gradient split
// (conv0_b16.hip); the default.  DX_CONV0_F32=1 selects the fp32-MFMA kernels above.
int launch_conv0_fwd_b16(const Conv0Args &a, const uint16_t *Wb, hipStream_t stream);
int launch_conv0_wgrad_b16(const Conv0Args &a, int nblocks, hipStream_t stream);
// round 6 (conv0_wgrad_ks.hip): 84 x 84 x 4 uint8 frames, the pixel contraction split over the waves, ONE slab per
// workgroup (conv0_wgrad_ks_workgroups(B) <= 256 of them); DX_CONV0_KS=0 keeps the tile kernel above
bool conv0_wgrad_ks_on();
bool conv0_wgrad_ks_supported(int in_h, int in_w, int in_c, int h0, int w0);
int conv0_wgrad_ks_workgroups(long long B);
int launch_conv0_wgrad_ks(const Conv0Args &a, int nblocks, hipStream_t stream);
// rollout-sized batches: 32x32 tile per workgroup, pre-split weight planes Wb [3][32][256] bf16
int launch_conv0_lat_b16(const Conv0Args &a, const uint16_t *Wb, hipStream_t stream);
// the rollout's whole conv stack -- optionally the whole act step, or T steps against the synthetic device
// env -- in ONE launch, one workgroup per 84 x 84 x 4 uint8 frame (convstack.hip): y0 / y1 stay in LDS
struct ConvStackArgs {
  uint8_t *obs;          // frames of this launch's envs at step 0: (B, 84, 84, 4) uint8; with `env` the whole rollout
                         // buffer (T + 1, row_stride envs, 84, 84, 4) from this launch's first env on: steps 1 .. T are WRITTEN
  const uint16_t *Wb0;   // conv0 weights, three bf16 planes [3][32][256] (k = (kh, kw, c))
  const float *bias0;
  const uint16_t *Wf1;   // conv1 weights in FRAGMENT order (launch_convstack_pack): [wave][8 steps][3 planes][64 lanes][8]
  const float *bias1;
  const uint16_t *Wf2;   // conv2 weights in fragment order: [wave][9 steps][3 planes][64 lanes][8]
  const float *bias2;
  float *y2;             // (B, 7, 7, 64) NHWC, or NULL when the tail runs in here
  int B;
  // ---- the policy's tail in the same kernel (NULL Wc: the caller runs its own): out = y2 Wc^T + beff, sampling ----
  const float *Wc, *beff;  // Wc (rows 0 .. A - 1 policy, row A value, the rest ZERO) in fragment order
                           // (launch_convstack_pack): [wave][2 tiles][8 rows][64 lanes][4]; beff [8]
  int A;
  const float *uniforms;   // optional (B): the uniforms to sample with (T = 1)
  uint64_t seed, counter;  // else uniform01(seed, counter + t, env0 + e)
  int env0;                // position of env 0 of this launch in the whole batch
  int64_t *actions;        // (T, row_stride): rows of this launch's envs
  float *log_prob, *values;
  // ---- T steps against the synthetic device env (synth_dev.hpp) in ONE launch: every env's chain
  // frame -> policy -> sample -> next frame is local to its workgroup ----
  int env, T, row_stride;
  float *rewards;
  uint8_t *resets;
  uint64_t env_seed, env_counter;
  float p_reward, p_reset;
  // ---- the forward of a TRAINING minibatch: `train` = 1, B images (obs[sample_idx[i]] when sample_idx is given), every
  // workgroup takes images blockIdx, blockIdx + grid, ...; y0 / y1 (fp32 NHWC, kept for the backward) are stored too ----
  int train;
  const int32_t *sample_idx;
  float *y0, *y1;              // (B, 20, 20, 32), (B, 9, 9, 64)
  int stamp_step;              // DX_DIAG only: the step whose phases are stamped
  unsigned long long *stamps;  // DX_DIAG only (DX_CS_DIAG=1): [B][8] shader-clock stamps of wave 0 (step 0), else NULL
};
bool convstack_supported(int in_h, int in_w, int in_c);
int launch_convstack(const ConvStackArgs &args, hipStream_t stream);
// the training forward (args.train) on `blocks` workgroups, waves specialised by layer, two images in flight (convstack_train.hip)
int launch_convstack_train(const ConvStackArgs &args, int blocks, hipStream_t stream);
// The conv-stack kernel's copies of conv1 / conv2's bf16 planes ([3][64][512], [3][64][576]) and of the
// factored tail's Wc ([8][3136], may be NULL) in the order its waves read them: every fragment load of the
// kernel is then ONE contiguous KB per wave (the planes' own layout made each a gather of 16 x 64 bytes).
long long convstack_pack_elems(int which);  // 0 / 1: uint16 elements of Wf1 / Wf2, 2: floats of the Wc copy
int launch_convstack_pack(const uint16_t *Wb1, const uint16_t *Wb2, const float *Wc, uint16_t *Wf1, uint16_t *Wf2, float *Wcf,
                          hipStream_t stream);
// the rollout's linear layer, weight-stationary split-K (fc_rollout.hip): slabs [parts][M][512], bias on slab 0
int fc_rollout_parts();
bool fc_rollout_supported(int M, int N, int K);
int launch_fc_rollout(const float *A, const float *W, const float *bias, float *slabs, int M, hipStream_t stream);
// the factored tail: linear layer + heads as one affine map of y2 (tail.hip, heads.hip)
bool tail_supported(int flat, int num_actions);  // 84 x 84 frames' conv output (3136 wide) and up to 18 actions
int tail_rows(int num_actions);                   // the A + 1 outputs padded to groups of eight (8, 16 or 24 rows)
long long tail_pack_scratch_floats(int num_actions);
long long tail_slab_floats(int B, int num_actions);
long long tail_slab_capacity_floats(int max_batch, int num_actions);  // >= tail_slab_floats of every B <= max_batch
// (`direct`, optional: the conv layers' bf16 planes of pack_direct_dev.hpp written by extra workgroups of the same launch)
struct TailDirectPlanes { uint16_t *p0, *f1, *f2, *d1, *d2; };
int launch_tail_pack(const float *params, const long long *off_w, const long long *off_b, int A, float *Wc, float *beff,
                     float *scratch, float *Wcf, hipStream_t stream, const TailDirectPlanes *direct = nullptr);  // Wcf (optional): Wc in convstack.hip's fragment order
int launch_tail_loss(const float *y2, const float *Wc, const float *beff, const int64_t *actions,
                     const float *old_log_prob, const float *advantages, const float *old_values,
                     const float *value_targets, const double *stats, float norm_eps, float *adv_norm_out, float *head,
                     float *dhead, int B, int A, int mode, float cliprange, float value_loss_coef, float entropy_coef,
                     long long global_batch, double *partials, int partials_capacity, unsigned *counter,
                     float *loss_out, hipStream_t stream);
int launch_tail_bwd(const float *y2, const float *dhead, const float *Wc, float *dy2, float *scratch, int B, int A,
                    hipStream_t stream);
// both in one pass over y2 (heads.hip: tail_loss_bwd_kernel; up to 7 actions, DX_ENOSUP beyond): the slab layout inside
// `scratch` is launch_tail_bwd's -- tail_bwd_plan (tail.hip) names it
struct TailBwdPlan { float *gslab, *sslab; int Jp, nwg, rows_per_wg; };
TailBwdPlan tail_bwd_plan(float *scratch, int B, int A);
int launch_tail_loss_bwd(const float *y2, const float *Wc, const float *beff, const int64_t *actions,
                         const float *old_log_prob, const float *advantages, const float *old_values,
                         const float *value_targets, const double *stats, float norm_eps, float *adv_norm_out, float *head,
                         float *dhead, int B, int A, int mode, float cliprange, float value_loss_coef, float entropy_coef,
                         long long global_batch, double *partials, int partials_capacity, unsigned *counter,
                         float *loss_out, float *dy2, float *gslab, float *sslab, int Jp, int nwg, int rows_per_wg,
                         hipStream_t stream);
int launch_tail_grads(const float *params, float *grads, const long long *off_w, const long long *off_b, int A,
                      float *scratch, int B, hipStream_t stream, bool reduced = false);
// the tail's G / s reduction in the same launch as the conv layers' slab reduction (two independent slab readers)
struct TailGreduceArgs;
TailGreduceArgs tail_greduce_args(float *scratch, int B, int A);
int launch_permute_reduce_greduce(const PermuteJob *jobs, int njobs, const TailGreduceArgs &g, hipStream_t stream);
int launch_tail_act(const float *y2, const float *Wc, const float *beff, int B, int A, const float *uniforms, uint64_t seed,
                    uint64_t counter, int64_t *actions, float *log_prob, float *values, hipStream_t stream);
int launch_tail_act_synth(const float *y2, const float *Wc, const float *beff, int B, int A, uint64_t seed, uint64_t counter,
                          int64_t *actions, float *log_prob, float *values, void *frames, long long frame_bytes,
                          float *rewards, uint8_t *resets, uint64_t env_seed, uint64_t env_counter, float p_reward,
                          float p_reset, int env0, long long vec0, hipStream_t stream);
// heads forward + categorical loss + heads dgrad / wgrad partials + loss scalars in one launch (heads.hip)
int launch_heads_loss_fused(const float *hid, const float *Wh, const float *bh, const int64_t *actions,
                            const float *old_log_prob, const float *advantages, const float *old_values,
                            const float *value_targets, const double *stats, float norm_eps, float *adv_norm_out,
                            float *head, float *dhead, float *dhid, float *slab, float *bias_slab, int nslab,
                            int rows_per_slab, int B, int A, int mode, float cliprange, float value_loss_coef,
                            float entropy_coef, long long global_batch, double *partials, int partials_capacity,
                            unsigned *counter, float *loss_out, hipStream_t stream);
int launch_heads_act_fused(const float *hid_slabs, int nslab, long long slab_stride, const float *Wh,
                           const float *bh, int B, int A, const float *uniforms, uint64_t seed,
                           uint64_t counter, int64_t *actions, float *log_prob, float *values,
                           hipStream_t stream);

// the rollout step's last launch against the synthetic device env: heads + sampling and the
// next observation batch / rewards / resets in ONE grid (heads.hip)
int launch_heads_act_synth(const float *hid_slabs, int nslab, long long slab_stride, const float *Wh,
                           const float *bh, int B, int A, uint64_t seed, uint64_t counter, int64_t *actions,
                           float *log_prob, float *values, void *frames, long long frame_bytes, float *rewards,
                           uint8_t *resets, uint64_t env_seed, uint64_t env_counter, float p_reward,
                           float p_reset, int env0, long long vec0, hipStream_t stream);
// (env0, vec0): position of this launch's first env / first 16-byte frame vector in the whole
// batch, so that a slice of the envs draws exactly what the whole-batch launch draws for it

}  // namespace dx

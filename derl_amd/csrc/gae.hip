// GAE backward scan as a wave-level segmented affine scan (gfx950, wave64).
//
// Replaces the Python loop of derl/runners/trajectory_transforms.py:56-62.  Each step is
// the affine map adv_t = delta_t + c_t * adv_{t+1} with c_t = (1-reset_t)*gamma*lambda;
// affine maps compose associatively, so T is cut into chunks of TC steps: every wave
// scans one chunk with carry 0 keeping (a_j, P_j) = (local advantage, product of c from
// step j to the chunk end) in registers, publishes its chunk aggregate (a_0, P_0) to LDS,
// folds the aggregates of the chunks after it into its carry-in and finishes
// adv_j = a_j + P_j * carry.  Lanes cover VEC consecutive envs each, so every global
// access is a coalesced row segment (VEC*4 B per lane); each input byte is read from HBM
// once (the v_{t+1} row at a chunk seam is re-read through L2) and each output written
// once: 17 B per element.
#include "common.hpp"

namespace {

template <int VEC> struct VecT;
template <> struct VecT<1> { using f = float; using u = uint8_t; };
template <> struct VecT<2> { using f = float2; using u = uint16_t; };
template <> struct VecT<4> { using f = float4; using u = uint32_t; };

template <int VEC>
__device__ __forceinline__ void load_f(const float *p, float (&out)[VEC]) {
  typename VecT<VEC>::f v = *reinterpret_cast<const typename VecT<VEC>::f *>(p);
  const float *s = reinterpret_cast<const float *>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) out[i] = s[i];
}

template <int VEC>
__device__ __forceinline__ void store_f(float *p, const float (&in)[VEC]) {
  typename VecT<VEC>::f v;
  float *s = reinterpret_cast<float *>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = in[i];
  *reinterpret_cast<typename VecT<VEC>::f *>(p) = v;
}

template <int VEC>
__device__ __forceinline__ void load_reset(const uint8_t *p, float (&not_reset)[VEC]) {
  typename VecT<VEC>::u v = *reinterpret_cast<const typename VecT<VEC>::u *>(p);
#pragma unroll
  for (int i = 0; i < VEC; ++i) not_reset[i] = ((v >> (8 * i)) & 0xff) ? 0.f : 1.f;
}

// block = 64 * W threads: lane -> VEC envs, wave -> chunk of TC timesteps.
template <int VEC, int TC, int MAXW>
__global__ __launch_bounds__(64 * MAXW) void gae_scan_kernel(
    const float *__restrict__ rewards, const uint8_t *__restrict__ resets,
    const float *__restrict__ values, const float *__restrict__ last_values, int T, int N,
    float gamma, float gamma_lambda, float *__restrict__ advantages,
    float *__restrict__ value_targets) {
  __shared__ float agg[2][MAXW][2][64 * VEC];  // [parity][wave][a0|P0][env]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int W = blockDim.x >> 6;
  const long long env0 = (static_cast<long long>(blockIdx.x) * 64 + lane) * VEC;
  const bool active = env0 < N;  // N % VEC == 0 is checked on the host
  const int span = W * TC;
  const int nsuper = (T + span - 1) / span;

  float carry_super[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) carry_super[e] = 0.f;

  for (int s = nsuper - 1; s >= 0; --s) {
    const int t0 = s * span + wave * TC;
    float a[TC][VEC], P[TC][VEC], v[TC][VEC];
    // ---- loads for the whole chunk first (memory-level parallelism), then the scan
    float r[TC][VEC], nr[TC][VEC], vnext_last[VEC];
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int t = t0 + j;
      if (active && t < T) {
        const size_t off = static_cast<size_t>(t) * N + env0;
        load_f<VEC>(rewards + off, r[j]);
        load_f<VEC>(values + off, v[j]);
        load_reset<VEC>(resets + off, nr[j]);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { r[j][e] = 0.f; v[j][e] = 0.f; nr[j][e] = 1.f; }
      }
    }
    {
      const int t = t0 + TC;  // first step of the next chunk (or the bootstrap value)
      if (active && t0 < T) {
        if (t < T) load_f<VEC>(values + static_cast<size_t>(t) * N + env0, vnext_last);
        else load_f<VEC>(last_values + env0, vnext_last);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) vnext_last[e] = 0.f;
      }
    }
    // ---- local backward scan with carry 0
    float acc[VEC], prod[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[e] = 0.f; prod[e] = 1.f; }
#pragma unroll
    for (int j = TC - 1; j >= 0; --j) {
      const int t = t0 + j;
      const bool valid = t < T;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        // v_{t+1}: next row of this chunk, the seam row, or last_values when t == T-1
        float vnext;
        if (j == TC - 1) vnext = vnext_last[e];
        else vnext = (t + 1 < T) ? v[j + 1][e] : vnext_last[e];
        if (valid) {
          const float g = nr[j][e] * gamma;
          const float delta = r[j][e] + g * vnext - v[j][e];
          const float c = nr[j][e] * gamma_lambda;
          acc[e] = delta + c * acc[e];
          prod[e] = c * prod[e];
        }
        a[j][e] = acc[e];
        P[j][e] = prod[e];
      }
    }
    // ---- publish the chunk aggregate, fold the later chunks into the carry
    const int par = s & 1;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      agg[par][wave][0][lane * VEC + e] = acc[e];
      agg[par][wave][1][lane * VEC + e] = prod[e];
    }
    __syncthreads();
    float carry[VEC], carry_next[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) carry_next[e] = carry_super[e];
    for (int w = W - 1; w >= 0; --w) {
      if (w == wave) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) carry[e] = carry_next[e];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        carry_next[e] = agg[par][w][0][lane * VEC + e] +
                        agg[par][w][1][lane * VEC + e] * carry_next[e];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) carry_super[e] = carry_next[e];
    // ---- finish and store
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int t = t0 + j;
      if (active && t < T) {
        float adv[VEC], vt[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          adv[e] = a[j][e] + P[j][e] * carry[e];
          vt[e] = adv[e] + v[j][e];
        }
        const size_t off = static_cast<size_t>(t) * N + env0;
        store_f<VEC>(advantages + off, adv);
        store_f<VEC>(value_targets + off, vt);
      }
    }
    // the next super-chunk writes agg[par ^ 1]; two barriers separate reuse of a parity
  }
}

template <int VEC, int TC, int MAXW>
int launch(const float *rewards, const uint8_t *resets, const float *values,
           const float *last_values, int T, int N, float gamma, float lambda, float *adv,
           float *vt, hipStream_t stream) {
  int W = dx::cdiv(T, TC);
  if (W > MAXW) W = MAXW;
  const int blocks = dx::cdiv(N, 64 * VEC);
  hipLaunchKernelGGL((gae_scan_kernel<VEC, TC, MAXW>), dim3(blocks), dim3(64 * W), 0, stream,
                     rewards, resets, values, last_values, T, N, gamma, gamma * lambda, adv, vt);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

extern "C" int dx_gae_f32(const float *rewards, const uint8_t *resets, const float *values,
                          const float *last_values, int T, int N, float gamma, float lambda,
                          float *advantages, float *value_targets, void *stream) {
  DX_TRACE("dx_gae_f32");
  DX_REQUIRE(T >= 0 && N >= 0, "dx_gae_f32: negative shape T=%d N=%d", T, N);
  if (T == 0 || N == 0) return DX_OK;
  DX_REQUIRE(rewards && resets && values && last_values && advantages && value_targets,
             "dx_gae_f32: null pointer");
  DX_REQUIRE(static_cast<long long>(T) * N < (1LL << 40), "dx_gae_f32: T*N too large");
  hipStream_t s = dx::as_stream(stream);
  const bool al16 = dx::aligned(rewards, 16) && dx::aligned(values, 16) &&
                    dx::aligned(last_values, 16) && dx::aligned(advantages, 16) &&
                    dx::aligned(value_targets, 16) && dx::aligned(resets, 4);
  // widest lane vector that still leaves >= 2 blocks per CU (256 CUs).  Measured at T=128,
  // N=2^20 (TB/s algorithmic): <4,4,16> 5.16, <4,8,8> 4.82, <4,8,4> 4.99, <4,16,4> 4.86,
  // <2,8,16> 4.47 -- short chunks on many waves win (90 VGPRs, 5 waves/SIMD).
  if (al16 && N % 4 == 0 && N >= 4 * 64 * 512)
    return launch<4, 4, 16>(rewards, resets, values, last_values, T, N, gamma, lambda,
                            advantages, value_targets, s);
  if (al16 && N % 2 == 0 && N >= 2 * 64 * 512)
    return launch<2, 8, 16>(rewards, resets, values, last_values, T, N, gamma, lambda,
                            advantages, value_targets, s);
  return launch<1, 8, 16>(rewards, resets, values, last_values, T, N, gamma, lambda,
                          advantages, value_targets, s);
}

// Counter-based synthetic environment step on the device (measurement input):
// uint8 frames i.i.d. uniform 0..255, rewards sign*[u < p_reward] in {-1,0,1}, resets
// Bernoulli(p_reset).  Stateless: every value is a hash of (seed, counter, position), so a
// step is one HBM-write-bound launch and reruns reproduce.
#include "common.hpp"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_atari_kernel(uint4 *frames, long long nvec, float *rewards,
                                                          uint8_t *resets, int nenvs, uint64_t seed,
                                                          uint64_t counter, float p_reward,
                                                          float p_reset) {
  const uint64_t key = mix64(seed * 0x9E3779B97F4A7C15ull + counter);
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const uint64_t a = mix64(key + 2 * static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ull);
    const uint64_t b = mix64(key + (2 * static_cast<uint64_t>(i) + 1) * 0x9E3779B97F4A7C15ull);
    frames[i] = make_uint4(static_cast<uint32_t>(a), static_cast<uint32_t>(a >> 32),
                           static_cast<uint32_t>(b), static_cast<uint32_t>(b >> 32));
  }
  const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid < nenvs) {
    const uint64_t r = mix64(~key + static_cast<uint64_t>(gid) * 0xD1B54A32D192ED03ull);
    const float u0 = static_cast<float>(r & 0xffffff) * (1.0f / 16777216.0f);
    const float u1 = static_cast<float>((r >> 24) & 0xffffff) * (1.0f / 16777216.0f);
    if (rewards) rewards[gid] = u0 < p_reward ? ((r >> 63) ? -1.f : 1.f) : 0.f;
    if (resets) resets[gid] = u1 < p_reset ? 1 : 0;
  }
}

}  // namespace

extern "C" int dx_synth_atari_step(void *frames, long long frame_bytes_total, float *rewards,
                                   uint8_t *resets, int nenvs, uint64_t seed, uint64_t counter,
                                   float p_reward, float p_reset, void *stream) {
  DX_REQUIRE(frames && frame_bytes_total > 0 && frame_bytes_total % 16 == 0 && dx::aligned(frames, 16),
             "dx_synth_atari_step: frames must be 16-byte aligned, size a multiple of 16");
  DX_REQUIRE(nenvs >= 0, "dx_synth_atari_step: nenvs < 0");
  const long long nvec = frame_bytes_total / 16;
  long long blocks = (nvec + 256 * 4 - 1) / (256 * 4);
  if (blocks > 4096) blocks = 4096;
  const long long need = (static_cast<long long>(nenvs) + 255) / 256;
  if (blocks < need) blocks = need;
  hipLaunchKernelGGL(synth_atari_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     dx::as_stream(stream), static_cast<uint4 *>(frames), nvec, rewards, resets, nenvs,
                     seed, counter, p_reward, p_reset);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

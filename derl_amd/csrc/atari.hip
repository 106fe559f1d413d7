// Atari frame pipeline of a batched env on the device:
//   MaxBetweenFrames (derl/env/atari_wrappers.py:121-137): obs = max(frame, previous frame)
//   ImagePreprocessing (:95-118): RGB -> gray, resize to 84 x 84            [parity UNPINNED]
//   QueueFrames (:140-163): the last K frames stacked (or concatenated) on the last axis
// with the auto-reset of derl's env batches (env_batch.py:66-70): an env that reports `done`
// restarts from its reset frame -- previous frame := reset frame, queue := K copies of it.
//
// All three are byte streams (HBM-bound): one thread per 4 output bytes, fully coalesced.
// max and queue are exact by construction and pinned to vectors recorded from the reference's
// own classes (tests/golden/atari_frames.npz).  The gray / resize step restates what the
// reference's cv2 calls compute in the default build -- BT.601 luma in 14-bit fixed point and a
// bilinear filter (the reference passes cv2.INTER_AREA in the `dst` position of cv2.resize, so
// the default INTER_LINEAR applies) -- but cv2 is absent from this image: NOT pinned.
#include "common.hpp"

namespace {

// out = max(raw, last); last := done ? reset : raw           (bytes; 16 per thread)
__global__ __launch_bounds__(256) void frame_max_kernel(const uint8_t *__restrict__ raw, uint8_t *__restrict__ last,
                                                       const uint8_t *__restrict__ dones,
                                                       const uint8_t *__restrict__ reset,
                                                       uint8_t *__restrict__ out, long long per_env,
                                                       long long total) {
  const long long i = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  const int n = static_cast<int>(i / per_env);  // per_env % 4 == 0: a thread never straddles envs
  const uint32_t r = *reinterpret_cast<const uint32_t *>(raw + i);
  const uint32_t l = *reinterpret_cast<const uint32_t *>(last + i);
  uint32_t m = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint32_t x = (r >> (8 * b)) & 0xff, y = (l >> (8 * b)) & 0xff;
    m |= (x > y ? x : y) << (8 * b);
  }
  *reinterpret_cast<uint32_t *>(out + i) = m;
  const bool done = dones != nullptr && dones[n] != 0;
  *reinterpret_cast<uint32_t *>(last + i) = done ? *reinterpret_cast<const uint32_t *>(reset + i) : r;
}

// One thread per frame element e = (pixel, channel) of one env: K output bytes.
//   stack : out[(e) * K + k]                 concat: out[pixel * C * K + k * C + c]
//   out[.., k] = k < K - 1 ? prev[.., k + 1] : frame[e];   done: every k = reset[e]
__global__ __launch_bounds__(256) void frame_queue_kernel(const uint8_t *__restrict__ prev,
                                                         const uint8_t *__restrict__ frame,
                                                         const uint8_t *__restrict__ dones,
                                                         const uint8_t *__restrict__ reset,
                                                         uint8_t *__restrict__ out, int N, long long E, int C,
                                                         int K, int concat) {
  const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= static_cast<long long>(N) * E) return;
  const int n = static_cast<int>(t / E);
  const long long e = t - static_cast<long long>(n) * E;
  const long long base = static_cast<long long>(n) * E * K +
                         (concat ? (e / C) * (static_cast<long long>(C) * K) + e % C : e * K);
  const long long sk = concat ? C : 1;
  const bool done = dones != nullptr && dones[n] != 0;
  const uint8_t newest = done ? reset[t] : frame[t];
  if (K == 4 && !concat) {  // the Nature-DQN shape: one 32-bit word per element
    const uint32_t p = *reinterpret_cast<const uint32_t *>(prev + base);
    const uint32_t shifted = (p >> 8) | (static_cast<uint32_t>(newest) << 24);
    *reinterpret_cast<uint32_t *>(out + base) = done ? newest * 0x01010101u : shifted;
    return;
  }
  for (int k = 0; k < K; ++k) {
    const uint8_t v = (done || k == K - 1) ? newest : prev[base + (k + 1) * sk];
    out[base + k * sk] = v;
  }
}

// gray (optional) + bilinear resize: one thread per output pixel (and channel when not gray).
// Luma: (R 4899 + G 9617 + B 1868 + 8192) >> 14.  Resize: source coordinate (d + 0.5) * scale -
// 0.5 clamped to the image, float32 lerp, round half up.
__global__ __launch_bounds__(256) void gray_resize_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                         int N, int H, int W, int C, int OH, int OW, int gray) {
  const int OC = gray ? 1 : C;
  const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const long long per = static_cast<long long>(OH) * OW * OC;
  if (t >= N * per) return;
  const int n = static_cast<int>(t / per);
  const int rem = static_cast<int>(t - n * per);
  const int c = rem % OC, ox = (rem / OC) % OW, oy = rem / (OC * OW);
  const float fy = fminf(fmaxf(__builtin_fmaf(oy + 0.5f, static_cast<float>(H) / OH, -0.5f), 0.f), H - 1.f);
  const float fx = fminf(fmaxf(__builtin_fmaf(ox + 0.5f, static_cast<float>(W) / OW, -0.5f), 0.f), W - 1.f);
  const int y0 = static_cast<int>(fy), x0 = static_cast<int>(fx);
  const int y1 = y0 + 1 < H ? y0 + 1 : y0, x1 = x0 + 1 < W ? x0 + 1 : x0;
  const float wy = fy - y0, wx = fx - x0;
  const uint8_t *img = in + static_cast<long long>(n) * H * W * C;
  auto sample = [&](int y, int x) -> float {
    const uint8_t *p = img + (static_cast<long long>(y) * W + x) * C;
    if (gray && C == 3) return static_cast<float>((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);
    return static_cast<float>(p[gray ? 0 : c]);
  };
  // explicit fused multiply-adds (one rounding each): the oracle restates exactly these
  const float top = __builtin_fmaf(wx, sample(y0, x1) - sample(y0, x0), sample(y0, x0));
  const float bot = __builtin_fmaf(wx, sample(y1, x1) - sample(y1, x0), sample(y1, x0));
  const float v = __builtin_fmaf(wy, bot - top, top);
  out[t] = static_cast<uint8_t>(fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f));
}

}  // namespace

extern "C" {

int dx_frame_max_u8(const uint8_t *raw, uint8_t *last, const uint8_t *dones, const uint8_t *reset,
                    uint8_t *out, int N, long long per_env, void *stream) {
  DX_TRACE("dx_frame_max_u8");
  DX_REQUIRE(raw && last && out && N >= 1 && per_env >= 4 && per_env % 4 == 0,
             "dx_frame_max_u8: bad arguments (N=%d, bytes per env=%lld: a multiple of 4)", N, per_env);
  DX_REQUIRE((dones == nullptr) == (reset == nullptr), "dx_frame_max_u8: dones and reset frames come together");
  DX_REQUIRE(dx::aligned(raw, 4) && dx::aligned(last, 4) && dx::aligned(out, 4) && (!reset || dx::aligned(reset, 4)),
             "dx_frame_max_u8: buffers must be 4-byte aligned");
  const long long total = per_env * N;
  hipLaunchKernelGGL(frame_max_kernel, dim3(dx::cdiv(total / 4, 256)), dim3(256), 0, dx::as_stream(stream), raw,
                     last, dones, reset, out, per_env, total);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int dx_frame_queue_u8(const uint8_t *prev, const uint8_t *frame, const uint8_t *dones, const uint8_t *reset,
                      uint8_t *out, int N, long long elems, int C, int K, int concat, void *stream) {
  DX_TRACE("dx_frame_queue_u8");
  DX_REQUIRE(prev && frame && out && N >= 1 && elems >= 1 && C >= 1 && K >= 1 && elems % C == 0,
             "dx_frame_queue_u8: bad arguments (N=%d, elements per frame=%lld, C=%d, K=%d)", N, elems, C, K);
  DX_REQUIRE((dones == nullptr) == (reset == nullptr), "dx_frame_queue_u8: dones and reset frames come together");
  DX_REQUIRE(prev != out, "dx_frame_queue_u8: the queue is shifted out of place (prev != out)");
  DX_REQUIRE(K != 4 || concat || (dx::aligned(prev, 4) && dx::aligned(out, 4)),
             "dx_frame_queue_u8: 4-frame stacks must be 4-byte aligned");
  hipLaunchKernelGGL(frame_queue_kernel, dim3(dx::cdiv(static_cast<long long>(N) * elems, 256)), dim3(256), 0,
                     dx::as_stream(stream), prev, frame, dones, reset, out, N, elems, C, K, concat);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int dx_gray_resize_u8(const uint8_t *in, uint8_t *out, int N, int H, int W, int C, int OH, int OW, int gray,
                      void *stream) {
  DX_TRACE("dx_gray_resize_u8");
  DX_REQUIRE(in && out && N >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1 && (C == 1 || C == 3),
             "dx_gray_resize_u8: bad arguments (N=%d, %dx%dx%d -> %dx%d)", N, H, W, C, OH, OW);
  const long long total = static_cast<long long>(N) * OH * OW * (gray ? 1 : C);
  hipLaunchKernelGGL(gray_resize_kernel, dim3(dx::cdiv(total, 256)), dim3(256), 0, dx::as_stream(stream), in, out,
                     N, H, W, C, OH, OW, gray);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // extern "C"

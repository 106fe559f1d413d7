// HBM-bound elementwise / reduction kernels of the update step (gfx950):
//   advantage normalisation  derl/runners/trajectory_transforms.py:89-92
//   global grad norm + clip  derl/alg/common.py:59-60 (torch clip_grad_norm_)
//   Adam / RMSprop steps     derl/factory/ppo.py:78-81, derl/factory/a2c.py:68-73
//   row gather by index      derl/runners/onpolicy.py:44-49,59-62
// All are one pass over their operands with 16-B lane accesses where alignment allows.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum of doubles; result valid in every thread
__device__ double block_sum(double v, double *scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += scratch[w];
  return s;
}

// stats[0] = sum, stats[1] = sum of squares, stats[2] = count (float64).  One block: the
// minibatch is a few thousand elements and the result feeds one tiny all-reduce when the
// batch is sharded over GPUs.
__global__ __launch_bounds__(1024) void adv_stats_kernel(const float *x, long long n, double *stats) {
  __shared__ double scratch[16];
  double s = 0.0, ss = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = x[i];
    s += v;
    ss += v * v;
  }
  s = block_sum(s, scratch);
  ss = block_sum(ss, scratch);
  if (threadIdx.x == 0) {
    stats[0] = s;
    stats[1] = ss;
    stats[2] = static_cast<double>(n);
  }
}

// The same statistics for every minibatch of a rollout at once: block s sums the elements
// x[index[s*seglen ..]] (a minibatch is a contiguous slice of an epoch's composed permutation) in
// the order adv_stats_kernel sums the gathered minibatch, so the doubles are bit-identical; a
// sharded run then needs ONE small all-reduce per rollout instead of one per minibatch.
__global__ __launch_bounds__(1024) void adv_stats_segments_kernel(const float *x, const int *index, long long n,
                                                                  long long seglen, double *stats) {
  __shared__ double scratch[16];
  const long long begin = blockIdx.x * seglen;
  const long long count = n - begin < seglen ? n - begin : seglen;
  double s = 0.0, ss = 0.0;
  for (long long i = threadIdx.x; i < count; i += blockDim.x) {
    const double v = x[index ? index[begin + i] : begin + i];
    s += v;
    ss += v * v;
  }
  s = block_sum(s, scratch);
  ss = block_sum(ss, scratch);
  if (threadIdx.x == 0) {
    double *o = stats + 3LL * blockIdx.x;
    o[0] = s;
    o[1] = ss;
    o[2] = static_cast<double>(count);
  }
}

// out = (x - mean) / (std + eps): population std from the float64 sums, applied in float32
// exactly like the NumPy expression (float32 array ops with float32 scalars).
__global__ __launch_bounds__(kThreads) void adv_apply_kernel(const float *x, float *out, long long n,
                                                             const double *stats, float eps) {
  const double cnt = stats[2];
  const double mean = stats[0] / cnt;
  double var = stats[1] / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = static_cast<float>(mean);
  const float denom = static_cast<float>(sqrt(var)) + eps;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (x[i] - meanf) / denom;
}

// adv_stats_kernel + adv_apply_kernel in ONE single-block launch for minibatch-sized inputs (the
// same sums in the same order, the same float32 expression: bit-identical outputs), one
// dependent launch less per update.
__global__ __launch_bounds__(1024) void adv_normalize_small_kernel(const float *x, float *out, long long n,
                                                                   double *stats, float eps) {
  __shared__ double scratch[16];
  __shared__ double total[2];
  double s = 0.0, ss = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = x[i];
    s += v;
    ss += v * v;
  }
  s = block_sum(s, scratch);
  ss = block_sum(ss, scratch);
  if (threadIdx.x == 0) {
    stats[0] = s;
    stats[1] = ss;
    stats[2] = static_cast<double>(n);
    total[0] = s;
    total[1] = ss;
  }
  __syncthreads();
  const double cnt = static_cast<double>(n);
  const double mean = total[0] / cnt;
  double var = total[1] / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = static_cast<float>(mean);
  const float denom = static_cast<float>(sqrt(var)) + eps;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) out[i] = (x[i] - meanf) / denom;
}

__global__ __launch_bounds__(kThreads) void sumsq_kernel(const float *g, long long n, double *partials) {
  __shared__ double scratch[4];
  double s = 0.0;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  const long long n4 = n / 4;
  const float4 *g4 = reinterpret_cast<const float4 *>(g);
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = g4[i];
    s += static_cast<double>(v.x) * v.x + static_cast<double>(v.y) * v.y +
         static_cast<double>(v.z) * v.z + static_cast<double>(v.w) * v.w;
  }
  for (long long i = n4 * 4 + static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    s += static_cast<double>(g[i]) * g[i];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// clip coefficient of torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6))
__device__ __forceinline__ float clip_coef(const double *partials, int npartials, float max_norm,
                                           float *norm_out) {
  // wave 0 sums the partials (lane-strided, then an xor butterfly: every workgroup forms the
  // same sum in the same order) and publishes the coefficient to the workgroup
  __shared__ float coef_shared;
  if (threadIdx.x < 64) {
    double s = 0.0;
    for (int i = threadIdx.x; i < npartials; i += 64) s += partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) {
      const float norm = static_cast<float>(sqrt(s));
      if (norm_out && blockIdx.x == 0) *norm_out = norm;
      float c = 1.f;
      if (max_norm > 0.f) {
        c = max_norm / (norm + 1e-6f);
        c = c < 1.f ? c : 1.f;
      }
      coef_shared = c;
    }
  }
  __syncthreads();
  return coef_shared;
}

struct OptArgs {
  float *p, *g, *m, *v;
  long long n;
  const double *partials;
  int npartials;
  float max_norm;
  float *norm_out;
  float lr_or_step_size, beta1, beta2, eps, bc2_sqrt;
  float one_minus_beta1, one_minus_beta2;  // formed in double on the host, like torch
  int vec4;
};

// Adam (torch semantics, no weight decay / amsgrad):
//   m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
// g is the clipped gradient and is written back (clip_grad_norm_ clips in place).
__device__ __forceinline__ void adam_one(const OptArgs &a, float coef, float &p, float &g, float &m, float &v) {
  g *= coef;
  m = m * a.beta1 + g * a.one_minus_beta1;
  v = v * a.beta2 + (g * g) * a.one_minus_beta2;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p - a.lr_or_step_size * (m / denom);
}

__global__ __launch_bounds__(kThreads) void adam_kernel(const OptArgs a) {
  const float coef = clip_coef(a.partials, a.npartials, a.max_norm, a.norm_out);
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long n4 = a.vec4 ? a.n / 4 : 0;  // 16-byte lanes when all four buffers are aligned
  for (long long i = tid; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4 *>(a.p)[i], g = reinterpret_cast<float4 *>(a.g)[i];
    float4 m = reinterpret_cast<float4 *>(a.m)[i], v = reinterpret_cast<float4 *>(a.v)[i];
    adam_one(a, coef, p.x, g.x, m.x, v.x);
    adam_one(a, coef, p.y, g.y, m.y, v.y);
    adam_one(a, coef, p.z, g.z, m.z, v.z);
    adam_one(a, coef, p.w, g.w, m.w, v.w);
    reinterpret_cast<float4 *>(a.p)[i] = p; reinterpret_cast<float4 *>(a.g)[i] = g;
    reinterpret_cast<float4 *>(a.m)[i] = m; reinterpret_cast<float4 *>(a.v)[i] = v;
  }
  for (long long i = 4 * n4 + tid; i < a.n; i += stride) {
    float p = a.p[i], g = a.g[i], m = a.m[i], v = a.v[i];
    adam_one(a, coef, p, g, m, v);
    a.p[i] = p; a.g[i] = g; a.m[i] = m; a.v[i] = v;
  }
}

// RMSprop (no momentum, not centered): s = alpha*s + (1-alpha)*g*g; p -= lr * g / (sqrt(s) + eps)
__global__ __launch_bounds__(kThreads) void rmsprop_kernel(const OptArgs a) {
  const float coef = clip_coef(a.partials, a.npartials, a.max_norm, a.norm_out);
  const float oma = a.one_minus_beta2;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const float g = a.g[i] * coef;
    const float s = a.v[i] * a.beta2 + (g * g) * oma;
    a.p[i] = a.p[i] - a.lr_or_step_size * (g / (sqrtf(s) + a.eps));
    a.v[i] = s;
    a.g[i] = g;
  }
}

// dst[i][:] = src[idx[i]][:], rows of row_bytes bytes; 16-B lanes when rows allow it.
template <typename V>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const uint8_t *src, const int32_t *idx,
                                                               uint8_t *dst, long long nrows,
                                                               long long row_bytes) {
  const long long per_row = row_bytes / static_cast<long long>(sizeof(V));
  const long long total = nrows * per_row;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / per_row, c = i - r * per_row;
    reinterpret_cast<V *>(dst + r * row_bytes)[c] =
        reinterpret_cast<const V *>(src + static_cast<long long>(idx[r]) * row_bytes)[c];
  }
}

// Up to kMaxGather arrays gathered by ONE index vector in one launch (blockIdx.y = array):
// the minibatch selection of every small per-sample array of a rollout.
constexpr int kMaxGather = 16;
struct GatherTable {
  const uint8_t *src[kMaxGather];
  uint8_t *dst[kMaxGather];
  int row_bytes[kMaxGather];
};

__global__ __launch_bounds__(kThreads) void gather_rows_multi_kernel(const GatherTable t, const int32_t *idx,
                                                                     long long nrows) {
  const uint8_t *src = t.src[blockIdx.y];
  uint8_t *dst = t.dst[blockIdx.y];
  const int rb = t.row_bytes[blockIdx.y];
  const int unit = (rb % 4 == 0) ? 4 : 1;  // host checked 4-byte alignment when rb % 4 == 0
  const long long per_row = rb / unit, total = nrows * per_row;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / per_row, c = i - r * per_row;
    const long long so = static_cast<long long>(idx[r]) * rb, d = r * rb;
    if (unit == 4)
      reinterpret_cast<uint32_t *>(dst + d)[c] = reinterpret_cast<const uint32_t *>(src + so)[c];
    else
      dst[d + c] = src[so + c];
  }
}

int grid_for(long long n, int per_thread) {
  long long b = (n + static_cast<long long>(kThreads) * per_thread - 1) / (static_cast<long long>(kThreads) * per_thread);
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return static_cast<int>(b);
}

}  // namespace

extern "C" int dx_adv_normalize_f32(const float *advantages, float *out, long long n, float eps,
                                    double *stats, int stats_ready, void *stream) {
  DX_TRACE("dx_adv_normalize_f32");
  DX_REQUIRE(n >= 0, "dx_adv_normalize_f32: negative n");
  if (n == 0) return DX_OK;
  DX_REQUIRE(advantages && out && stats, "dx_adv_normalize_f32: null pointer");
  hipStream_t s = dx::as_stream(stream);
  if (!stats_ready && n <= 16384 && advantages != out) {  // minibatch-sized: one launch
    hipLaunchKernelGGL(adv_normalize_small_kernel, dim3(1), dim3(1024), 0, s, advantages, out, n, stats, eps);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
  if (!stats_ready) {
    hipLaunchKernelGGL(adv_stats_kernel, dim3(1), dim3(1024), 0, s, advantages, n, stats);
    DX_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(adv_apply_kernel, dim3(grid_for(n, 4)), dim3(kThreads), 0, s, advantages, out, n,
                     stats, eps);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_adv_stats_f32(const float *advantages, long long n, double *stats, void *stream) {
  DX_TRACE("dx_adv_stats_f32");
  DX_REQUIRE(n > 0 && advantages && stats, "dx_adv_stats_f32: bad argument");
  hipLaunchKernelGGL(adv_stats_kernel, dim3(1), dim3(1024), 0, dx::as_stream(stream), advantages, n, stats);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_adv_stats_segments_f32(const float *advantages, const int32_t *index, long long n,
                                         long long seglen, double *stats, void *stream) {
  DX_TRACE("dx_adv_stats_segments_f32");
  DX_REQUIRE(n > 0 && seglen > 0 && advantages && stats, "dx_adv_stats_segments_f32: bad argument");
  const long long nseg = (n + seglen - 1) / seglen;
  DX_REQUIRE(nseg <= 65535, "dx_adv_stats_segments_f32: %lld segments (max 65535)", nseg);
  hipLaunchKernelGGL(adv_stats_segments_kernel, dim3(static_cast<unsigned>(nseg)), dim3(1024), 0,
                     dx::as_stream(stream), advantages, index, n, seglen, stats);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_grad_sumsq_f32(const float *grads, long long n, double *partials, int npartials,
                                 void *stream) {
  DX_TRACE("dx_grad_sumsq_f32");
  DX_REQUIRE(n > 0 && grads && partials, "dx_grad_sumsq_f32: bad argument");
  DX_REQUIRE(npartials >= 1 && npartials <= 4096, "dx_grad_sumsq_f32: npartials=%d out of [1,4096]", npartials);
  DX_REQUIRE(dx::aligned(grads, 16), "dx_grad_sumsq_f32: grads must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(npartials), dim3(kThreads), 0, dx::as_stream(stream), grads, n,
                     partials);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

static int check_opt(const char *who, float *p, float *g, float *v, long long n, const double *partials,
                     int npartials) {
  DX_REQUIRE(n > 0 && p && g && v, "%s: bad argument", who);
  DX_REQUIRE(npartials >= 0 && npartials <= 4096 && (npartials == 0 || partials),
             "%s: bad partials (n=%d)", who, npartials);
  return DX_OK;
}

extern "C" int dx_clip_adam_step_f32(float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                                     long long n, const double *sumsq_partials, int npartials,
                                     double max_norm, double lr, double beta1, double beta2,
                                     double eps, long long step, float *norm_out, void *stream) {
  DX_TRACE("dx_clip_adam_step_f32");
  if (int rc = check_opt("dx_clip_adam_step_f32", params, grads, exp_avg_sq, n, sumsq_partials, npartials))
    return rc;
  DX_REQUIRE(exp_avg && step >= 1, "dx_clip_adam_step_f32: exp_avg null or step < 1");
  // hyper-parameters arrive as doubles (torch keeps them as Python floats) and every derived
  // scalar is formed in double before the single rounding to float32
  const double bc1 = 1.0 - pow(beta1, static_cast<double>(step));
  const double bc2 = 1.0 - pow(beta2, static_cast<double>(step));
  OptArgs a{params, grads, exp_avg, exp_avg_sq, n, sumsq_partials, npartials,
            static_cast<float>(max_norm), norm_out, static_cast<float>(lr / bc1),
            static_cast<float>(beta1), static_cast<float>(beta2), static_cast<float>(eps),
            static_cast<float>(sqrt(bc2)), static_cast<float>(1.0 - beta1),
            static_cast<float>(1.0 - beta2),
            dx::aligned(params, 16) && dx::aligned(grads, 16) && dx::aligned(exp_avg, 16) &&
                    dx::aligned(exp_avg_sq, 16) ? 1 : 0};
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 8)), dim3(kThreads), 0, dx::as_stream(stream), a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_clip_rmsprop_step_f32(float *params, float *grads, float *square_avg, long long n,
                                        const double *sumsq_partials, int npartials, double max_norm,
                                        double lr, double alpha, double eps, float *norm_out,
                                        void *stream) {
  DX_TRACE("dx_clip_rmsprop_step_f32");
  if (int rc = check_opt("dx_clip_rmsprop_step_f32", params, grads, square_avg, n, sumsq_partials, npartials))
    return rc;
  OptArgs a{params, grads, nullptr, square_avg, n, sumsq_partials, npartials,
            static_cast<float>(max_norm), norm_out, static_cast<float>(lr), 0.f,
            static_cast<float>(alpha), static_cast<float>(eps), 1.f, 1.f,
            static_cast<float>(1.0 - alpha), 0};
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n, 4)), dim3(kThreads), 0, dx::as_stream(stream), a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_gather_rows_multi(const void *const *src, void *const *dst, const long long *row_bytes,
                                    int narrays, const int32_t *idx, long long nrows, void *stream) {
  DX_TRACE("dx_gather_rows_multi");
  DX_REQUIRE(narrays >= 0 && narrays <= kMaxGather, "dx_gather_rows_multi: %d arrays (max %d)", narrays,
             kMaxGather);
  DX_REQUIRE(nrows >= 0, "dx_gather_rows_multi: negative row count");
  if (narrays == 0 || nrows == 0) return DX_OK;
  DX_REQUIRE(src && dst && row_bytes && idx, "dx_gather_rows_multi: null pointer");
  GatherTable t;
  long long biggest = 0;
  for (int i = 0; i < narrays; ++i) {
    DX_REQUIRE(src[i] && dst[i] && row_bytes[i] > 0 && row_bytes[i] < (1 << 30),
               "dx_gather_rows_multi: bad array %d", i);
    t.src[i] = static_cast<const uint8_t *>(src[i]);
    t.dst[i] = static_cast<uint8_t *>(dst[i]);
    t.row_bytes[i] = static_cast<int>(row_bytes[i]);
    if (row_bytes[i] % 4 == 0)
      DX_REQUIRE(dx::aligned(src[i], 4) && dx::aligned(dst[i], 4), "dx_gather_rows_multi: array %d misaligned", i);
    const long long units = nrows * (row_bytes[i] % 4 == 0 ? row_bytes[i] / 4 : row_bytes[i]);
    if (units > biggest) biggest = units;
  }
  hipLaunchKernelGGL(gather_rows_multi_kernel, dim3(grid_for(biggest, 4), narrays), dim3(kThreads), 0,
                     dx::as_stream(stream), t, idx, nrows);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_gather_rows(const void *src, const int32_t *idx, void *dst, long long nrows,
                              long long row_bytes, void *stream) {
  DX_TRACE("dx_gather_rows");
  DX_REQUIRE(nrows >= 0 && row_bytes > 0, "dx_gather_rows: bad shape");
  if (nrows == 0) return DX_OK;
  DX_REQUIRE(src && idx && dst, "dx_gather_rows: null pointer");
  hipStream_t s = dx::as_stream(stream);
  const uint8_t *sp = static_cast<const uint8_t *>(src);
  uint8_t *dp = static_cast<uint8_t *>(dst);
  if (row_bytes % 16 == 0 && dx::aligned(src, 16) && dx::aligned(dst, 16))
    hipLaunchKernelGGL(gather_rows_kernel<uint4>, dim3(grid_for(nrows * (row_bytes / 16), 4)),
                       dim3(kThreads), 0, s, sp, idx, dp, nrows, row_bytes);
  else if (row_bytes % 4 == 0 && dx::aligned(src, 4) && dx::aligned(dst, 4))
    hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(grid_for(nrows * (row_bytes / 4), 4)),
                       dim3(kThreads), 0, s, sp, idx, dp, nrows, row_bytes);
  else
    hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, dim3(grid_for(nrows * row_bytes, 4)),
                       dim3(kThreads), 0, s, sp, idx, dp, nrows, row_bytes);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

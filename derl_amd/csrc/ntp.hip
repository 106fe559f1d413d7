// Persistent NT GEMM for the 64-column conv stages (conv1 / conv2 forward, conv2 dgrad):
// C[m][n] = epi(sum_k A(m,k) * Wp[n][k]) with the implicit A matrix of igemm.hpp.
//
// The LDS-tiled kernels of igemm.hip / igemm_pix.hip start a workgroup per output tile: with K =
// 512 / 576 (and 64..576 for a dgrad pixel, whose invalid taps are skipped) a tile is 2..18 K steps,
// and every tile pays its own memory round trip before the first MFMA and its own epilogue after
// the last.  Here 512 workgroups (two per CU) stay resident and walk their tiles as ONE stream of
// K steps:
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces of 8 rows x 128 bytes) in a
//     ring of three stages; the fills for step g + 2 are issued behind the barrier of step g, so
//     the ring keeps running across tile boundaries: the next tile's first operands are in
//     flight while this tile's last MFMAs and its epilogue run;
//   * ONE barrier per K step behind a counted vmcnt (loads, stores and LDS-DMA complete in issue
//     order: the two steps after an epilogue count its 32 stores);
//   * a 128 x 64 tile on 4 waves of 64 x 32: a lane reads 4 consecutive k of its row with one
//     ds_read_b128 (3 reads feed 8 MFMAs); the DMA writes LDS linearly, so the bank swizzle is
//     applied to the SOURCE address (16-byte chunk c of row R sits in slot c ^ ((R >> 1) & 7));
//   * MODE 0 (forward): a tile is 128 consecutive output pixels; MODE 1 (dgrad): a tile is one
//     input pixel of 128 images, so the set of valid taps is uniform and the others are skipped;
//     the workgroups of an XCD walk the pixels of the same image groups (shared taps from L2).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "igemm_dev.hpp"

namespace dx {
namespace {

using f4 = __attribute__((ext_vector_type(4))) float;

constexpr int kBM = 128, kBN = 64, kBK = 32, kRing = 3;
constexpr int kStageFloats = (kBM + kBN) * kBK;  // 6144 floats = 24 KiB
constexpr int kDma = 6;                          // LDS-DMA instructions per wave per stage

struct NtpArgs {
  NTArgs nt;
  int ntiles;         // MODE 0: row tiles
  int nimg, ngroups;  // MODE 1: images, groups of 128 images
  int tiles_per_xcd;  // MODE 1: ceil(ngroups / 8) * pixels
  int TA, TB, PA, PB;  // run grid and the element pitch of a step in ta / tb (see Cursor)
  int diag;           // DX_NTP_DIAG bits: 1 = stamps, 2 = cycles at the wait + barrier; WRONG RESULTS: 4 = no fills
                      // after the first two, 8 = no barrier, 16 = no epilogue loads / stores, 32 = every tile reads the rows of the first
};

// Position in the stream of K steps: tile `i` of this workgroup, tap (ta, tb), offset q inside the
// tap's run.  The runs of a row form a TA x TB grid (forward: TA = kernel rows, TB = 1; dgrad: one
// run per kernel tap, tap (ta, tb) reads pixel (y - ta, x - tb)), so the runs that lie inside the
// image are a rectangle [ta_lo, ta_hi] x [tb_lo, tb_hi] and every offset is arithmetic: no table
// look-ups (scalar loads) on the per-step path.
struct Cursor {
  int i;        // tile counter of this workgroup
  int valid;    // tile exists
  int tile;     // MODE 0: row tile; MODE 1: image group
  int pix;      // MODE 1: pixel
  int pixoff;   // MODE 1: element offset of the pixel inside an image
  int ta, tb, q;
  int ta_hi, tb_lo, tb_hi;
};

template <int MODE>
__device__ __forceinline__ void open_tile(Cursor &c, const NtpArgs &p) {
  const Gather &g = p.nt.g;
  if (MODE == 0) {
    c.tile = blockIdx.x + c.i * gridDim.x;
    c.valid = c.tile < p.ntiles;
    c.pix = 0; c.pixoff = 0;
    c.ta = 0; c.ta_hi = p.TA - 1; c.tb_lo = 0; c.tb_hi = 0;
  } else {
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const uint32_t u = (blockIdx.x >> 3) + c.i * per;
    const uint32_t gl = fdiv(u, g.div_img);
    c.pix = u - gl * g.OHW;
    c.tile = gl * 8 + xcd;
    c.valid = u < static_cast<uint32_t>(p.tiles_per_xcd) && c.tile < p.ngroups;
    const int y = fdiv(c.pix, g.div_row), x = c.pix - y * g.OW;  // unit stride (checked on the host)
    c.ta = max(0, y - g.H + 1); c.ta_hi = min(p.TA - 1, y);
    c.tb_lo = max(0, x - g.W + 1); c.tb_hi = min(p.TB - 1, x);
    c.pixoff = (y * g.W + x) * g.C;
  }
  c.tb = c.tb_lo;
  c.q = 0;
}

// -> true when the step just left was the last of its tile
template <int MODE>
__device__ __forceinline__ bool advance(Cursor &c, const NtpArgs &p) {
  c.q += kBK;
  if (c.q < p.nt.g.seglen) return false;
  c.q = 0;
  if (++c.tb <= c.tb_hi) return false;
  c.tb = c.tb_lo;
  if (++c.ta <= c.ta_hi) return false;
  ++c.i;
  open_tile<MODE>(c, p);
  return true;
}

template <int MODE, int EPI>
__global__ __launch_bounds__(512, 2) void ntp_kernel(const NtpArgs p, unsigned long long *stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned long long t_entry = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
  const NTArgs &a = p.nt;
  const Gather &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3;  // index inside its role
  Cursor first;
  first.i = 0;
  open_tile<MODE>(first, p);
  if (!first.valid) return;  // uniform: this workgroup has no tile

  if (wave8 >= 4) {
    // ================= loader waves: fills only =================
    // wave w fills A pieces 4w .. 4w+3 (rows 32w .. 32w+31) and W pieces 2w, 2w+1 of every stage
    const int lrow = lane >> 3;
    const float *wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = (2 * wave + j) * 8 + lrow;
      wp[j] = a.Wp + static_cast<long long>(n) * a.K + 4 * ((lane & 7) ^ ((n >> 1) & 7));
    }
    const float *ap[4];
    Cursor ld = first;  // where the NEXT fill goes
#define DX_NTP_ROWS()                                                                            \
  _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                             \
    const int R = 32 * wave + 8 * q_ + lrow;                                                     \
    const int chunk = (lane & 7) ^ ((R >> 1) & 7);                                               \
    long long base;                                                                              \
    if (MODE == 0) {                                                                             \
      const uint32_t m = ld.tile * kBM + R;                                                      \
      const uint32_t img = fdiv(m, g.div_img), rem = m - img * g.OHW;                            \
      const uint32_t oy = fdiv(rem, g.div_row), ox = rem - oy * g.OW;                            \
      base = static_cast<long long>(img) * g.img_stride + (oy * g.sy * g.W + ox * g.sx) * g.C;   \
    } else {                                                                                     \
      base = static_cast<long long>(ld.tile * kBM + R) * g.img_stride;                           \
    }                                                                                            \
    ap[q_] = static_cast<const float *>(g.src) + base + 4 * chunk;                               \
  }
#define DX_NTP_FILL(SLOT)                                                                        \
  {                                                                                              \
    float *dst_ = smem + (SLOT) * kStageFloats;                                                  \
    const int ao_ = ld.pixoff + ld.ta * p.PA + ld.tb * p.PB + ld.q;                              \
    const int ko_ = (ld.ta * p.TB + ld.tb) * g.seglen + ld.q;                                    \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                             \
        __builtin_amdgcn_global_load_lds(ap[q_] + ao_, dst_ + (4 * wave + q_) * 256, 16, 0, 0); \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                             \
        __builtin_amdgcn_global_load_lds(wp[j_] + ko_, dst_ + (16 + 2 * wave + j_) * 256, 16, 0, 0); \
    const int before_ = ld.tile;                                                                 \
    if (advance<MODE>(ld, p) && ld.valid && (MODE == 0 || ld.tile != before_)) { DX_NTP_ROWS() } \
    ++issued;                                                                                    \
  }
    DX_NTP_ROWS()
    int issued = 0, slot = 2;  // slot of the next fill
    DX_NTP_FILL(0)
    if (ld.valid) DX_NTP_FILL(1)
    unsigned long long l_vm = 0, l_bar = 0, l_all = stamps ? __builtin_amdgcn_s_memtime() : 0;
    for (int gstep = 0; gstep < issued; ++gstep) {
      // hand step gstep over: its fill is the oldest in flight, at most one younger fill behind it
      const unsigned long long l0 = (stamps && (p.diag & 2)) ? __builtin_amdgcn_s_memtime() : 0;
      if (issued - gstep >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long l1 = (stamps && (p.diag & 2)) ? __builtin_amdgcn_s_memtime() : 0;
      __builtin_amdgcn_s_barrier();  // consumers are done with step gstep - 1: its slot is free
      asm volatile("" ::: "memory");
      if (stamps && (p.diag & 2)) { l_vm += l1 - l0; l_bar += __builtin_amdgcn_s_memtime() - l1; }
      if (ld.valid && !(p.diag & 4)) {
        DX_NTP_FILL(slot)
        slot = slot == kRing - 1 ? 0 : slot + 1;
      } else if (ld.valid) {  // diagnostic: walk the cursor without filling
        advance<MODE>(ld, p);
        ++issued;
      }
    }
#undef DX_NTP_FILL
#undef DX_NTP_ROWS
    if (stamps && lane == 0) {  // loader stamps behind the consumers': total, vmcnt wait, barrier wait
      unsigned long long *o = stamps + static_cast<long long>(gridDim.x) * 4 * 7 + (static_cast<long long>(blockIdx.x) * 4 + wave) * 3;
      o[0] = __builtin_amdgcn_s_memtime() - l_all; o[1] = l_vm; o[2] = l_bar;
    }
    return;
  }

  // ================= consumer waves: fragment reads, MFMAs, epilogues =================
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves of 64 x 32
  const char *lds = reinterpret_cast<const char *>(smem);
  const unsigned x = (l31 >> 1) & 7;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned slot = ((2 * q + hi) ^ x) * 16;
    aoff[q] = (wm * 64 + l31) * 128 + slot;
    boff[q] = (kBM + wn * 32 + l31) * 128 + slot;
  }
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  Cursor cs = first;  // the step being multiplied
  int slot = 0;
  const int n = wn * 32 + l31;
  const float bias = (EPI == EPI_BIAS_RELU) ? a.bias[n] : 0.f;
  // a use of the bias in front of the loop: otherwise the compiler waits for this load at its
  // first use, inside the epilogue
  asm volatile("" ::"v"(bias));

  const unsigned long long t_loop = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
  const unsigned long long c_loop = stamps ? __builtin_amdgcn_s_memtime() : 0;
  unsigned long long c_wait = 0;
  int nsteps = 0, ntiles = 0;
  while (cs.valid) {
    const unsigned long long c_w0 = (stamps && (p.diag & 2)) ? __builtin_amdgcn_s_memtime() : 0;
    __builtin_amdgcn_s_barrier();  // the loaders arrive once this step's fill has landed
    asm volatile("" ::: "memory");
    if (stamps && (p.diag & 2)) c_wait += __builtin_amdgcn_s_memtime() - c_w0;
    ++nsteps;
    if (p.diag & 64) {  // diagnostic: loaders alone (no fragment reads, no MFMAs)
      slot = slot == kRing - 1 ? 0 : slot + 1;
      advance<MODE>(cs, p);
      continue;
    }
    const char *base = lds + slot * kStageFloats * 4;
    f4 af[2][2], bf[2];
    af[0][0] = *reinterpret_cast<const f4 *>(base + aoff[0]);
    af[0][1] = *reinterpret_cast<const f4 *>(base + aoff[0] + 32 * 128);
    bf[0] = *reinterpret_cast<const f4 *>(base + boff[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = q & 1, nx = c ^ 1;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0][0], bf[c][0], acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < 4) {  // the next group's fragments, requested in the first MFMA's shadow
        af[nx][0] = *reinterpret_cast<const f4 *>(base + aoff[q + 1]);
        af[nx][1] = *reinterpret_cast<const f4 *>(base + aoff[q + 1] + 32 * 128);
        bf[nx] = *reinterpret_cast<const f4 *>(base + boff[q + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1][0], bf[c][0], acc[1], 0, 0, 0);
#pragma unroll
      for (int e = 1; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0][e], bf[c][e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1][e], bf[c][e], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = slot == kRing - 1 ? 0 : slot + 1;

    const int tile = cs.tile, pix = cs.pix;
    if (!advance<MODE>(cs, p)) continue;
    // ---- epilogue of (tile, pix): C/D layout column = l31, row = (r & 3) + 8 (r >> 2) + 4 hi.
    // Tiles are never ragged (the launchers require whole tiles).
    const int row0 = tile * kBM + wm * 64 + 4 * hi;  // GEMM row (MODE 0) or image (MODE 1)
    if (p.diag & 16) {  // diagnostic: no epilogue traffic
    } else if (MODE == 0) {
      float *o = a.out + static_cast<long long>(row0) * a.ldc + n;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[static_cast<long long>(32 * t + (r & 3) + 8 * (r >> 2)) * a.ldc] = fmaxf(acc[t][r] + bias, 0.f);
    } else {
      const long long imgo = static_cast<long long>(g.OHW) * a.ldc;
      const long long o0 = static_cast<long long>(row0) * imgo + static_cast<long long>(pix) * a.ldc + n;
      float mk[2][16];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mk[t][r] = a.mask_src[o0 + (32 * t + (r & 3) + 8 * (r >> 2)) * imgo];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          a.out[o0 + (32 * t + (r & 3) + 8 * (r >> 2)) * imgo] = mk[t][r] > 0.f ? acc[t][r] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    ++ntiles;
  }
  if (stamps && lane == 0) {  // DX_NTP_DIAG: 100 MHz ticks (entry, loop start, exit), loop cycles, wait cycles, steps, tiles
    unsigned long long *o = stamps + (static_cast<long long>(blockIdx.x) * 4 + wave) * 7;
    o[0] = t_entry; o[1] = t_loop; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = __builtin_amdgcn_s_memtime() - c_loop;
    o[4] = c_wait; o[5] = nsteps; o[6] = ntiles;
  }
}

bool ntp_on() {  // DX_NTP=0: these stages on the per-tile kernels
  static int v = -1;
  if (v < 0) { const char *e = getenv("DX_NTP"); v = e ? atoi(e) : 1; }
  return v != 0;
}

int ntp_workgroups() {  // DX_NTP_NWG: resident workgroups (default 512 = two per CU)
  static int v = -1;
  if (v < 0) { const char *e = getenv("DX_NTP_NWG"); v = e ? atoi(e) : 512; }
  return v < 8 ? 8 : v / 8 * 8;
}

template <int MODE, int EPI>
int launch_as(const NtpArgs &p, hipStream_t stream) {
  constexpr int lds = kRing * kStageFloats * 4;
  static bool configured = false;
  if (!configured) {
    DX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ntp_kernel<MODE, EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  static const int diag = getenv("DX_NTP_DIAG") ? atoi(getenv("DX_NTP_DIAG")) : 0;
  const int grid = ntp_workgroups();
  if (!diag) {
    hipLaunchKernelGGL((ntp_kernel<MODE, EPI>), dim3(grid), dim3(512), lds, stream, p, nullptr);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
  // diagnostic: in-kernel stamps, summarised on stderr (synchronous; never on the product path)
  NtpArgs pd = p;
  pd.diag = diag;
  unsigned long long *dev = nullptr;
  const size_t count = static_cast<size_t>(grid) * 4 * 7, lcount = static_cast<size_t>(grid) * 4 * 3;
  DX_HIP(hipMalloc(&dev, (count + lcount) * 8));
  DX_HIP(hipMemsetAsync(dev, 0, (count + lcount) * 8, stream));
  hipLaunchKernelGGL((ntp_kernel<MODE, EPI>), dim3(grid), dim3(512), lds, stream, pd, dev);
  DX_LAUNCH_CHECK();
  DX_HIP(hipStreamSynchronize(stream));
  std::vector<unsigned long long> h(count + lcount);
  DX_HIP(hipMemcpy(h.data(), dev, (count + lcount) * 8, hipMemcpyDeviceToHost));
  DX_HIP(hipFree(dev));
  unsigned long long first = ~0ull, last = 0;
  std::vector<double> pro, loop, cps, wfrac, steps;
  for (size_t i = 0; i < count; i += 7) {
    if (h[i + 5] == 0) continue;
    first = std::min(first, h[i]); last = std::max(last, h[i + 2]);
    pro.push_back((h[i + 1] - h[i]) * 0.01); loop.push_back((h[i + 2] - h[i + 1]) * 0.01);
    cps.push_back(static_cast<double>(h[i + 3]) / h[i + 5]); wfrac.push_back(static_cast<double>(h[i + 4]) / h[i + 3]);
    steps.push_back(static_cast<double>(h[i + 5]));
  }
  std::vector<double> lvm, lbar;
  for (size_t i = count; i < count + lcount; i += 3) {
    if (h[i] == 0) continue;
    lvm.push_back(static_cast<double>(h[i + 1]) / h[i]); lbar.push_back(static_cast<double>(h[i + 2]) / h[i]);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mx = [](const std::vector<double> &v) { return *std::max_element(v.begin(), v.end()); };
  fprintf(stderr, "[ntp mode %d M=%d K=%d grid=%d] span %.1f us | per wave: prologue %.2f us, loop median %.1f max %.1f us, "
          "steps median %.0f max %.0f, %.0f cycles/step (ideal 4096), wait+barrier %.1f %% of the loop\n", MODE, p.nt.M, p.nt.K,
          grid, (last - first) * 0.01, med(pro), med(loop), mx(loop), med(steps), mx(steps), med(cps), 100 * med(wfrac));
  if (!lvm.empty())
    fprintf(stderr, "        loaders: %.1f %% of their time waiting for a fill to land, %.1f %% at the barrier\n", 100 * med(lvm), 100 * med(lbar));
  return DX_OK;
}

}  // namespace

// Forward conv stage (bias + ReLU).  DX_ENOSUP = not covered: the caller keeps its own kernel.
int launch_ntp_fwd(const NTArgs &a, hipStream_t stream) {
  const Gather &g = a.g;
  if (!ntp_on() || a.N != kBN || g.idx || g.check || a.om.enabled || a.ksplit != 1 || g.seglen % kBK ||
      g.nseg > kMaxSeg || a.K != g.nseg * g.seglen || a.M < kBM * 512 || a.M % kBM || a.ldc != kBN)
    return DX_ENOSUP;
  DX_REQUIRE(aligned(g.src, 16) && aligned(a.Wp, 16) && g.C % 4 == 0, "ntp: operands must be 16-byte aligned");
  NtpArgs p;
  p.nt = a;
  p.ntiles = cdiv(a.M, kBM);
  p.nimg = p.ngroups = p.tiles_per_xcd = p.diag = 0;
  p.TA = g.nseg; p.TB = 1; p.PA = g.nseg > 1 ? g.seg_off[1] : 0; p.PB = 0;
  for (int s2 = 0; s2 < g.nseg; ++s2)
    if (g.seg_off[s2] != s2 * p.PA) return DX_ENOSUP;
  return launch_as<0, EPI_BIAS_RELU>(p, stream);
}

// dgrad stage tiled as one pixel x 128 images (ReLU mask from the kept activation)
int launch_ntp_pix(const NTArgs &a, int nimg, int TA, int TB, hipStream_t stream) {
  const Gather &g = a.g;
  if (TA * TB != g.nseg || TB < 2 || g.sy != 1 || g.sx != 1 || !g.check) return DX_ENOSUP;
  for (int s2 = 0; s2 < g.nseg; ++s2)  // tap (ta, tb) reads pixel (y - ta, x - tb)
    if (g.seg_dy[s2] != -(s2 / TB) || g.seg_dx[s2] != -(s2 % TB) ||
        g.seg_off[s2] != (s2 / TB) * g.seg_off[TB] + (s2 % TB) * g.seg_off[1])
      return DX_ENOSUP;
  if (!ntp_on() || a.N != kBN || g.idx || a.om.enabled || a.ksplit != 1 || g.seglen % kBK || g.nseg > kMaxSeg ||
      a.K != g.nseg * g.seglen || nimg < kBM * 8 || nimg % kBM || static_cast<long long>(nimg) * g.OHW != a.M ||
      !a.mask_src || a.ldc != kBN)
    return DX_ENOSUP;
  DX_REQUIRE(aligned(g.src, 16) && aligned(a.Wp, 16) && g.C % 4 == 0, "ntp: operands must be 16-byte aligned");
  NtpArgs p;
  p.nt = a;
  p.ntiles = p.diag = 0;
  p.TA = TA; p.TB = TB; p.PA = g.seg_off[TB]; p.PB = g.seg_off[1];
  p.nimg = nimg;
  p.ngroups = cdiv(nimg, kBM);
  p.tiles_per_xcd = cdiv(p.ngroups, 8) * g.OHW;
  return launch_as<1, EPI_MASK>(p, stream);
}

}  // namespace dx

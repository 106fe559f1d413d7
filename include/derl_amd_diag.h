/* derl_amd_diag -- microbenchmark entry points (libderl_amd_diag.so, built from
 * derl_amd/csrc/experiments/diag.hip).  NOT part of the product boundary (include/derl_amd.h):
 * these kernels rebuild pieces of the GEMM loops in isolation to find out what the fp32 matrix
 * pipe of gfx950 sustains (tools/mfma_peak.py, tools/gemm_loop.py, tools/clock_probe.py).  Same
 * conventions as derl_amd.h: device pointers, hipStream_t as void*, 0 / DX_E* status codes. */
#ifndef DERL_AMD_DIAG_H
#define DERL_AMD_DIAG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: `blocks` workgroups of 4 waves, every wave issuing iters x 4 independent
 * v_mfma_f32_32x32x2_f32 with no memory traffic (blocks * 4 * iters * 4 * 4096 flop): what the
 * fp32 matrix pipe sustains on this part, timed by the caller (tools/mfma_peak.py). */
int dx_diag_mfma_f32(int blocks, int iters, float *out, void *stream);
/* The same flop count as one dependent accumulation chain per wave. */
int dx_diag_mfma_f32_chain(int blocks, int iters, float *out, void *stream);
/* The K loop of the 128x64 GEMM tile fed from LDS only (no global loads, LDS writes or barriers):
 * blocks x 4 waves x iters x 32 MFMAs; mode 0 reads fragments right before use, 1 one step ahead. */
int dx_diag_lds_mfma_f32(int blocks, int iters, int mode, float *out, void *stream);
/* The NT GEMM K loop rebuilt step by step on plain row-major operands (tools/gemm_loop.py). */
int dx_diag_gemm_loop_f32(const float *A, const float *B, int tiles, int ktiles, int what, float *out,
                          void *stream);

#ifdef __cplusplus
}
#endif

#endif /* DERL_AMD_DIAG_H */

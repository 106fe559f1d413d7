/* derl_amd -- C-ABI of the MI355X (gfx950) on-policy hot path.
 *
 * mknbv/derl has no FFI: its extension boundary is the Python API.  This
 * header is the boundary UNDER that API: each entry point replaces the inside of one
 * reference function (cited as file:line relative to the reference root) with
 * hand-written CDNA4 HIP kernels.  Host Python (the derl_amd Python package, same class / argument
 * names as derl) binds these through ctypes; INTEGRATION.md shows the stub a derl
 * maintainer would add.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer unless its name ends in _host;
 *  - the caller allocates and owns every buffer (including workspaces); the library keeps
 *    no pointer past a call;
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); calls
 *    enqueue work asynchronously and return without synchronising;
 *  - every function returns 0 on success, a negative DX_E* code on failure and never
 *    throws; dx_last_error() returns a thread-local message for the last failure;
 *  - shapes are validated on the host before any launch (a bad shape never reaches a
 *    kernel).
 */
#ifndef DERL_AMD_H
#define DERL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DX_OK 0
#define DX_EINVAL (-1)   /* bad argument (shape, null pointer, alignment)            */
#define DX_EHIP (-2)     /* a HIP runtime call failed                               */
#define DX_ENOSUP (-3)   /* configuration not supported by the compiled kernels     */
#define DX_EWS (-4)      /* workspace too small                                     */

#define DX_ABI_VERSION 1

int dx_abi_version(void);
const char *dx_last_error(void);
/* name_host: buffer of >= 256 bytes; cu_count/lds_bytes may be NULL. */
int dx_device_info(int device, char *name_host, int *cu_count, int *lds_bytes);

/* ---------------------------------------------------------------------------------
 * GAE backward scan -- replaces the loop of derl/runners/trajectory_transforms.py:45-65
 * (GAE.__call__).  Layout is the reference's np.asarray stacking: time-major (T, N)
 * row-major; `values` is the squeezed (T, N) view of the (T, N, 1) array, `resets` one
 * byte per element (numpy/torch bool), `last_values` (N) = value of
 * state["latest_observations"] (:47-50).
 *   delta_t = r_t + (1-reset_t)*gamma*v_{t+1} - v_t
 *   adv_t   = delta_t + (1-reset_t)*gamma*lambda*adv_{t+1},  adv_T = 0
 *   value_targets = adv + values                                   (:63)
 * Wave-level segmented affine scan: lanes cover envs (coalesced rows), waves cover
 * chunks of T, chunk aggregates are combined through LDS.  fp32; differs from the
 * reference's float64-intermediate evaluation by <= ~1e-6 (tests state the tolerance).
 * Algorithmic HBM traffic: 17 B per (t, n) element + 4 B per env.
 * --------------------------------------------------------------------------------- */
int dx_gae_f32(const float *rewards, const uint8_t *resets, const float *values,
               const float *last_values, int T, int N, float gamma, float lambda,
               float *advantages, float *value_targets, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DERL_AMD_H */

// Shared helpers for the gfx950 kernels of libbdm_hip.so.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BDM_OK 0
#define BDM_ERR_ARG 1
#define BDM_ERR_LAUNCH 2
#define BDM_ERR_UNSUPPORTED 3

namespace bdm {

// Records a message retrievable through bdm_last_error(); never exits the process
// (the reference's CUDA_CHECK_ERRORS calls exit(-1), cuda_utils.cuh:28-37).
void set_error(const char *fmt, ...);

// hipGetLastError() is per host thread and sticky: it also returns (and clears) an error left behind by an EARLIER
// runtime call of this thread that nobody checked (e.g. one of the host framework's).  The message says so, so that
// such an error is not mistaken for a fault of the kernel named here.
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s (HIP's per-thread last-error state: raised by this launch or left by an earlier unchecked call)",
              what, hipGetErrorString(e));
    return BDM_ERR_LAUNCH;
  }
  return BDM_OK;
}

#define BDM_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      bdm::set_error(__VA_ARGS__);  \
      return BDM_ERR_ARG;           \
    }                               \
  } while (0)

// Raise a kernel's dynamic-LDS ceiling (up to the CU's 160 KiB); not a stream operation.  The attribute is sticky per
// (kernel, device), and hipFuncSetAttribute is a slow driver call (measured ~0.2 ms: at one call per launch it made every
// big-LDS kernel host-bound), so each expansion site remembers the size it has already granted on each device.
// BDM_STAGING (read per call: the equality tests flip it): "0" keeps the global-memory / one-workgroup forms of the three kernels
// that have an LDS-staged form (devoxelisation gather, sparse feature gather, 32^3 voxel plan in slabs), "1" forces the staged
// form wherever it fits; unset = the measured per-shape choice.  -> -1 (unset), 0, 1
inline int bdm_staging_choice() {
  const char *e = getenv("BDM_STAGING");
  return (e == nullptr || e[0] == 0) ? -1 : (e[0] == '0' ? 0 : 1);
}

#define BDM_ALLOW_LDS(kernel, bytes)                                                                      \
  do {                                                                                                    \
    if ((bytes) > 48 * 1024) {                                                                            \
      static int granted_[16] = {0};                                                                      \
      int dev_ = 0;                                                                                       \
      (void)hipGetDevice(&dev_);                                                                          \
      if (dev_ < 0 || dev_ >= 16 || granted_[dev_] < (int)(bytes)) {                                      \
        hipError_t e_ = hipFuncSetAttribute((const void *)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)(bytes));                                                \
        if (e_ != hipSuccess) {                                                                           \
          bdm::set_error("%s: cannot raise the dynamic LDS limit to %d bytes: %s", #kernel, (int)(bytes), \
                         hipGetErrorString(e_));                                                          \
          return BDM_ERR_LAUNCH;                                                                          \
        }                                                                                                 \
        if (dev_ >= 0 && dev_ < 16) granted_[dev_] = (int)(bytes);                                        \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// Unfused IEEE multiply/add: the distance arithmetic shared with the oracle is defined
// without FMA contraction (oracle/pvcnn_ops_ref.c header).
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// 64-lane reductions on DPP row operations (xor 1, xor 2, half-row mirror, row mirror: four single-pass moves inside a row of 16
// lanes) + one readlane per row, rows combined in order 0..3: the result is wave-uniform.  The `__shfl_xor` butterflies these replace are
// six DEPENDENT ds_bpermute round trips through the LDS crossbar per value (twelve for a double) -- measured 5.3 us for the 32 dot products
// of one SE gate (round 5); every GroupNorm statistic, row mean and amax of the small kernels goes through one of these.  Fixed order:
// deterministic and independent of the batch.  All 64 lanes must be active.
#if defined(__HIPCC__)
#define BDM_DPP_QUAD_XOR1 0xB1
#define BDM_DPP_QUAD_XOR2 0x4E
#define BDM_DPP_ROW_HALF_MIRROR 0x141
#define BDM_DPP_ROW_MIRROR 0x140
__device__ __forceinline__ float dpp_f32(float v, const int ctrl) {
  switch (ctrl) {
    case BDM_DPP_QUAD_XOR1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), BDM_DPP_QUAD_XOR1, 0xF, 0xF, true));
    case BDM_DPP_QUAD_XOR2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), BDM_DPP_QUAD_XOR2, 0xF, 0xF, true));
    case BDM_DPP_ROW_HALF_MIRROR: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), BDM_DPP_ROW_HALF_MIRROR, 0xF, 0xF, true));
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), BDM_DPP_ROW_MIRROR, 0xF, 0xF, true));
  }
}
__device__ __forceinline__ double dpp_f64(double v, const int ctrl) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int lo2, hi2;
  switch (ctrl) {
    case BDM_DPP_QUAD_XOR1: lo2 = __builtin_amdgcn_update_dpp(0, lo, BDM_DPP_QUAD_XOR1, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, BDM_DPP_QUAD_XOR1, 0xF, 0xF, true); break;
    case BDM_DPP_QUAD_XOR2: lo2 = __builtin_amdgcn_update_dpp(0, lo, BDM_DPP_QUAD_XOR2, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, BDM_DPP_QUAD_XOR2, 0xF, 0xF, true); break;
    case BDM_DPP_ROW_HALF_MIRROR: lo2 = __builtin_amdgcn_update_dpp(0, lo, BDM_DPP_ROW_HALF_MIRROR, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, BDM_DPP_ROW_HALF_MIRROR, 0xF, 0xF, true); break;
    default: lo2 = __builtin_amdgcn_update_dpp(0, lo, BDM_DPP_ROW_MIRROR, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, BDM_DPP_ROW_MIRROR, 0xF, 0xF, true); break;
  }
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ float readlane_f32(float v, const int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double readlane_f64(double v, const int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// CONTRACT of every helper below: called by a FULL wave in converged control flow -- all 64 lanes active (block sizes are
// multiples of 64 and the call does not sit under a divergent branch).  The DPP steps run with bound_ctrl (an inactive source lane reads 0)
// but the row totals are fetched with v_readlane from lanes 0 / 16 / 32 / 48, whose registers are STALE when those lanes are inactive;
// debug builds (-DBDM_DEBUG_WAVES) trap on a partial wave.  wave_sum associates ((row0 + row1) + row2) + row3: it is NOT bit-compatible
// with the xor butterfly of rounds 1 - 4 (wave_sum_bfly, row16_sum and half32_sum are).
#ifdef BDM_DEBUG_WAVES
#define BDM_FULL_WAVE() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define BDM_FULL_WAVE() do { } while (0)
#endif
__device__ __forceinline__ float wave_sum(float v) {
  BDM_FULL_WAVE();
  v += dpp_f32(v, BDM_DPP_QUAD_XOR1);
  v += dpp_f32(v, BDM_DPP_QUAD_XOR2);
  v += dpp_f32(v, BDM_DPP_ROW_HALF_MIRROR);
  v += dpp_f32(v, BDM_DPP_ROW_MIRROR);
  return ((readlane_f32(v, 0) + readlane_f32(v, 16)) + readlane_f32(v, 32)) + readlane_f32(v, 48);
}
__device__ __forceinline__ double wave_sum(double v) {
  BDM_FULL_WAVE();
  v += dpp_f64(v, BDM_DPP_QUAD_XOR1);
  v += dpp_f64(v, BDM_DPP_QUAD_XOR2);
  v += dpp_f64(v, BDM_DPP_ROW_HALF_MIRROR);
  v += dpp_f64(v, BDM_DPP_ROW_MIRROR);
  return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}
// Sum over the 16 lanes of a DPP row, left in every lane of the row; over the 32 lanes of a half wave, left in every lane of the half.
// BIT-IDENTICAL to the `__shfl_xor` butterflies (1, 2, 4, 8[, 16]) they replace: after each step all lanes of a group hold the group's sum and
// the next step adds the same two partial sums (in the other order: floating-point addition is commutative).
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32(v, BDM_DPP_QUAD_XOR1);
  v += dpp_f32(v, BDM_DPP_QUAD_XOR2);
  v += dpp_f32(v, BDM_DPP_ROW_HALF_MIRROR);
  v += dpp_f32(v, BDM_DPP_ROW_MIRROR);
  return v;
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64(v, BDM_DPP_QUAD_XOR1);
  v += dpp_f64(v, BDM_DPP_QUAD_XOR2);
  v += dpp_f64(v, BDM_DPP_ROW_HALF_MIRROR);
  v += dpp_f64(v, BDM_DPP_ROW_MIRROR);
  return v;
}
__device__ __forceinline__ float half32_sum(float v) {   // all 64 lanes active
  BDM_FULL_WAVE();
  v = row16_sum(v);
  const float lo = readlane_f32(v, 0) + readlane_f32(v, 16), hi = readlane_f32(v, 32) + readlane_f32(v, 48);
  return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ double half32_sum(double v) {
  BDM_FULL_WAVE();
  v = row16_sum(v);
  const double lo = readlane_f64(v, 0) + readlane_f64(v, 16), hi = readlane_f64(v, 32) + readlane_f64(v, 48);
  return (threadIdx.x & 32) ? hi : lo;
}
// ... and over all 64 lanes in the butterfly's association ((row 0 + row 1) + (row 2 + row 3)): bit-identical to xor 1 .. 32; wave-uniform
__device__ __forceinline__ double wave_sum_bfly(double v) {
  BDM_FULL_WAVE();
  v = row16_sum(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  BDM_FULL_WAVE();
  v = fmaxf(v, dpp_f32(v, BDM_DPP_QUAD_XOR1));
  v = fmaxf(v, dpp_f32(v, BDM_DPP_QUAD_XOR2));
  v = fmaxf(v, dpp_f32(v, BDM_DPP_ROW_HALF_MIRROR));
  v = fmaxf(v, dpp_f32(v, BDM_DPP_ROW_MIRROR));
  return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}
#endif

// workspace layout of the deterministic voxeliser (pvcnn_ops.hip: vox_plan_kernel)
struct VoxWs {
  int *start;   // [b][r3]  exclusive prefix of the per-voxel counts
  int *tmp;     // [b][n]
  int *sorted;  // [b][n]   per-voxel point lists, ascending point index inside a voxel
};
static inline VoxWs vox_ws(void *ws, int b, int n, int r3) {
  VoxWs w;
  w.start = (int *)ws;
  w.tmp = w.start + (size_t)b * r3;
  w.sorted = w.tmp + (size_t)b * n;
  return w;
}

// Swish(x) = x sigmoid(x) as x * rcp(1 + 2^(-x log2 e)): five instructions (v_exp_f32 and v_rcp_f32 are 1 ulp each) instead of
// the ~20 of expf + an IEEE division -- this sits in the inner loop of every kernel that applies a folded GroupNorm + Swish.
// <= 3 ulp from the exactly rounded value; -inf..+inf behave (x -> -0 / x).
#if defined(__HIPCC__)
__device__ __forceinline__ float swishf(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.44269504088896340736f));
}
#endif

// LDS-return settle point.  Measured on MI355X / ROCm 7.2 (tools/two_proc_repro.hip): when a workgroup with
// heavy 16-byte LDS traffic (sparse_gemm_s3_kernel) is co-resident on the CU -- another stream or another process -- a VALU
// instruction issued right behind the compiler's partial wait `s_waitcnt lgkmcnt(N > 0)` can read a VGPR whose ds_read has NOT
// landed yet (ds_write x 3, ds_read_b64 x 2, s_waitcnt lgkmcnt(1), v_pk_fma: one operand stale in ~50 % of the launches; never
// when the kernel runs alone).  Waiting for ALL outstanding LDS operations before the first use removes it (0 / 240 launches).
// The kernels that consume LDS-resident parameters inside their staging code call this right after the read.
#if defined(__HIPCC__)
__device__ __forceinline__ void lds_settle(float &a, float &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lds_settle8(float (&v)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
#endif

}  // namespace bdm

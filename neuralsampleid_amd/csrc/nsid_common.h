// Shared device/host helpers for the gfx950 kernels. Internal header (the public C ABI is include/nsid.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nsid.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Activation storage: fp32 or bf16 (include/nsid.h NSID_F32 / NSID_BF16). Streaming kernels move one 16-byte chunk per
// lane per access in either type: 4 fp32 or 8 bf16; arithmetic is always fp32.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float* v) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = x[e];
  }
  __device__ static __forceinline__ void store(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  }
};
template <> struct Chunk<__bf16> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const __bf16* p, float* v) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
  }
  __device__ static __forceinline__ void store(__bf16* p, const float* v) {
    bf16x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (__bf16)v[e];      // RNE (v_cvt_pk_bf16_f32), NaN stays NaN
    *reinterpret_cast<bf16x8*>(p) = x;
  }
};

// N consecutive per-channel fp32 values (BatchNorm scale/shift/mean/...) as float4 loads; c is a multiple of 4
template <int N>
__device__ __forceinline__ void load_channels(const float* __restrict__ p, int c, float* v) {
#pragma unroll
  for (int e = 0; e < N; e += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(p + c + e);
    v[e] = x[0]; v[e + 1] = x[1]; v[e + 2] = x[2]; v[e + 3] = x[3];
  }
}

// dispatch a templated launch on the runtime storage code
#define NSID_DISPATCH_DTYPE(dtype, T, ...)                    \
  do {                                                        \
    if ((dtype) == NSID_BF16) { typedef __bf16 T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; }                    \
  } while (0)
#define NSID_DTYPE_OK(d) ((d) == NSID_F32 || (d) == NSID_BF16)

#define NSID_WAVE 64

// activation codes shared by every kernel (include/nsid.h: NSID_ACT_*)
__device__ __forceinline__ float nsid_act(float v, int act) {
  switch (act) {
    case NSID_ACT_RELU: return v < 0.f ? 0.f : v;      // NaN-propagating, as torch.relu
    case NSID_ACT_LEAKY: return v > 0.f ? v : 0.2f * v;
    case NSID_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}
// derivative of the activation, expressed on the pre-activation value
__device__ __forceinline__ float nsid_act_grad(float pre, int act) {
  switch (act) {
    case NSID_ACT_RELU: return pre > 0.f ? 1.f : 0.f;
    case NSID_ACT_LEAKY: return pre > 0.f ? 1.f : 0.2f;
    default: return 1.f;
  }
}

// 64-lane sum with DPP only (xor-1 / xor-2 in quads, mirrors within 8 and 16 lanes, row broadcasts 15 and 31) and a
// scalar read-back of lane 63: no LDS permutes on the critical path. Every lane receives the total.
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  // row_bcast adds lane 15 of each 16-lane row into the next row (rows 1,3), then lane 31 into rows 2,3; lanes outside
  // the row mask add the bound_ctrl zero
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, true));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// hipGetLastError() reports the last error of ANY earlier runtime call on this thread (torch's own included), so the
// error state is cleared right before a launch and read right after it.
#define NSID_LAUNCH(kernel, grid, block, shmem, stream, ...)                \
  do {                                                                      \
    (void)hipGetLastError();                                                \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);    \
  } while (0)

static inline int nsid_launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) fprintf(stderr, "[nsid] kernel launch failed: %s\n", hipGetErrorString(e));
  return e == hipSuccess ? NSID_OK : NSID_ELAUNCH;
}
static inline bool nsid_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define NSID_REQUIRE(cond) \
  do {                     \
    if (!(cond)) return NSID_EINVAL; \
  } while (0)

// wgrad.hip: 128x128-tile form of the bf16 weight-gradient GEMM; returns 1 when the shape is outside its preconditions
int nsid_wgrad2_launch(const void* dout, int ldd, const void* x, int ldx, float* dw, int M, int Nout, int K, int groups,
                       const float* in_scale, const float* in_shift, float slope, hipStream_t stream);

// gemm256.hip: forward GEMM on 256x256 tiles with LDS-DMA staging (bf16 activations and weights, one group); returns 1 when the
// shape is outside its preconditions
int nsid_gemm256_fwd_launch(const void* x, int ldx, const void* w, const float* bias, const void* addend, int ldadd, void* out,
                            int ldo, int M, int Nout, int K, bool relu_out, float* stat, long stat_plane, long stat_ld,
                            hipStream_t stream);
extern void* g_gemm_trace_host;        // gemm.hip: the buffer installed by nsid_debug_gemm_trace (nullptr = none)

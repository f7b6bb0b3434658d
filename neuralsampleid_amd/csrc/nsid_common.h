// Shared device/host helpers for the gfx950 kernels. Internal header (the public C ABI is include/nsid.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nsid.h"

// (Rounds 3-4 kept result-changing timing switches -- NSID_ABN_NOMATH / NOSIDE, NSID_WGRAD_PLAINSTORE, NSID_G256_ABLATE, NSID_F256_* --
// behind #ifdefs in the product kernels; their measurements are in docs/experiments.md and the switches are gone: the product sources
// compile one arithmetic. Timing experiments belong in tools/variants/.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
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

// ---- tuning table (include/nsid.h: nsid_set_tuning / nsid_get_tuning / nsid_reset_tuning; defined in tuning.hip) -----------
// Every launch heuristic that has a number in it reads that number from here. The defaults are the values that won their one-box
// A/B of the whole two-stream step (docs/experiments.md); the library never reads the environment, so the arithmetic and the kernel
// selection of a run depend only on what the caller sets explicitly (bench.py --tune key=value records it in its JSON line).
#define NSID_TUNING_TABLE(X)                                                                                                  \
  X(stream_max_wg, 2048)       /* workgroup cap of the grid-stride streaming kernels (bn_apply, ...) */                       \
  X(bn_bwd_apply_max_wg, 384)  /* BatchNorm-backward apply pass: 1.5 workgroups per CU leave room for the other view */      \
  X(gemm_deep_ks, 2)           /* 32-deep MFMA sub-steps per LDS stage for launches of <= gemm_deep_max_wg workgroups */      \
  X(gemm_deep_pd, 2)           /* register sets (= stages in flight) of those launches: 2 or 4 */                             \
  X(gemm_deep_max_wg, 1024)    /* round 5 re-sweep (weight-stationary kernels in): 512 -> 1024 = -0.8 % of the step (x3), deep plan neutral */ \
  X(gemm_deep_ec, 0)           /* early commit (three LDS stage buffers): wins alone, loses in the two-stream step */         \
  X(gemm_deep_kinds, 5)        /* which GEMM kinds take the deep form: bit 0 forward, bit 1 backward-data, bit 2 weight gradient */ \
  X(bwd_split_max_tiles, 64)   /* fp32-storage backward-data launches of at most this many tiles and Nout >= 1024 split their reduction (atomics); 0 = never */ \
  X(fwd_narrow, -1)            /* -1: shape heuristic; 0 / 1 force 128- / 64-wide forward tiles */                             \
  X(bwd_narrow, -1)                                                                                                           \
  X(g256_min, 512)             /* >= this many 256x256 tiles: gemm256.hip (LDS-DMA staging); 0 = never */                     \
  X(g256_train, 0)             /* 1: launches WITH a statistics epilogue (training) may take gemm256.hip too */               \
  X(g256_grid, 0)              /* workgroups of the persistent gemm256 launch (0 = one per CU) */                             \
  X(wgrad_wide, 1)             /* 8-wave 128x128-tile weight gradients (wgrad.hip) where they apply */                        \
  X(wgrad_rect, 1)             /* 128x64-tile weight gradients */                                                             \
  X(wgrad_wgs_rect, 256)       /* workgroup target of the 128x64 form (fewer splits = fewer atomic bytes) */                  \
  X(wgrad_wgs_sq, 1024)                                                                                                       \
  X(wgrad_rect_min, 256)                                                                                                      \
  X(wgrad_sq_min, 256)                                                                                                        \
  X(w3_wgs, 256)                                                                                                              \
  X(w3_min_tiles, 64)                                                                                                         \
  X(knn_strips, 0)             /* 1: always the general strip kernel (tests of the fallback) */                               \
  X(knn_pair_min, 512)         /* clips per launch from which knn2 runs as two 128-VGPR workgroups per CU; 0 = never */        \
  X(knn_raw16, 1)              /* forward-only launches (bf16 features, no affine, >= knn_pair_min clips): one bf16 MFMA pass on the raw features */ \
  X(knn_raw_wpe, 6)            /* its waves per SIMD (4 / 6 = two / three workgroups per CU) */ \
  X(knn_sel_min_n, 128)        /* graphs of at least this many nodes take the in-register threshold select for k*d > 8 */       \
  X(mr_grid_stride, 0)         /* 1: grid-stride aggregation instead of the LDS-staged per-clip kernel */                     \
  X(mr_split, 0)               /* channel split of the LDS-staged aggregation (0 = heuristic) */                              \
  X(ffn_regs, 3)               /* eval FFN with the x tile in registers (ffn256_fused.hip) also at C = 64 (bit 0) / C = 128 (bit 1) */ \
  X(mr_key_min_k, 8)           /* bf16 aggregation with >= this many neighbours: integer-key search (mr_fwd_key_kernel); 0 = never */ \
  X(mr_bwd_sorted_min_k, 3)    /* bf16 aggregation backward with >= this many neighbours: degree-ranked gather (mr_bwd_sorted_kernel); 0 = never */ \
  X(ffn_waves, 8)              /* waves per workgroup of the fused eval-mode FFN (4 or 8) */                                  \
  X(ffn256, 1)                 /* 1: the C = 256 stage's eval-mode FFN as one launch (ffn256_fused.hip); 0: two GEMM launches */ \
  X(mrconv_variant, 7)         /* fused eval-mode aggregation + grouped conv: bit 0 = 8 waves, bit 1 = direct 8-byte stores, bit 2 = one group per workgroup over a range of clips at C = 256 (bit 3: at every width) */ \
  X(mrconv_pg_wgs, 768)        /* workgroups of that form */ \
  X(wgg_rows, 4096)            /* rows per workgroup of the grouped (deferred) weight gradients: splits per view = M / wgg_rows (one-box A/B: 1024 / 2048 / 4096 / 8192 / 16384 rows: 7.67 / 7.47 / 7.42 / 7.56 / 7.88 ms) */ \
  X(wgg_rows_sq, 1024)         /* the same for its 64x64-tile class (the C = 64 layers: few tiles, long row loops) */ \
  X(wgg_rows_gen, 512)         /* and for its predicated class (stem, the 32-channel grouped conv) */ \
  X(bn_fin_tiles, 0)           /* BatchNorm finalize kernels: from this many row tiles on a workgroup covers 16 channels x 64 tile groups instead of 64 x 16 (0: never; 256 measured equal within noise, docs/experiments.md round 6) */ \
  X(wgg_w3, 1)                 /* grouped problems with Nout % 128 == 0 and K % 128 == 0 on 128x128 tiles, 8 waves (wgrad.hip) */ \
  X(ws_gemm, 7)                /* weight-stationary streaming GEMMs (wsgemm.hip) for the small-K layers: bit 0 forward, bit 1 backward-data, bit 2 backward-data with the BatchNorm backward on its operand load */

enum NsidTuneKey {
#define NSID_TUNE_ENUM(name, def) NSID_T_##name,
  NSID_TUNING_TABLE(NSID_TUNE_ENUM)
#undef NSID_TUNE_ENUM
  NSID_T_COUNT
};
extern long g_nsid_tune[NSID_T_COUNT];
static inline long nsid_tune(NsidTuneKey k) { return g_nsid_tune[k]; }

// ---- launch counters (include/nsid.h nsid_debug_counter): how many launches took each kernel VARIANT since the last reset. Host
// side, incremented when a launch is enqueued (also under stream capture). Tests use them to prove that the variants the bench
// times (full-tile, 64-deep stages, rectangular weight-gradient tiles, ...) are the ones a parity test exercised.
#define NSID_COUNTER_TABLE(X)                                                                  \
  X(gemm_fwd) X(gemm_bwd_data) X(gemm_bwd_weight)        /* launches of gemm.hip by kind */      \
  X(gemm_full)            /* predication-free full-tile instantiation */                        \
  X(gemm_ks2)             /* 64-deep LDS stages */                                              \
  X(gemm_pd4) X(gemm_ec)                                                \
  X(gemm_bwd_split)       /* backward-data with a split reduction (the projector head) */   \
  X(gemm_split_major)     /* split index fastest in the grid (a split stays on one XCD) */      \
  X(gemm_affine_load)     /* producer BatchNorm + activation applied on the operand load */     \
  X(gemm_relu_load)                                                                             \
  X(gemm_bn_sums)         /* backward-data epilogue emits BatchNorm-backward column sums */     \
  X(gemm_bn_apply_load)   /* backward-data applies a BatchNorm backward on its operand load */  \
  X(gemm256)              /* gemm256.hip */                                                     \
  X(ws_fwd) X(ws_bwd_data) X(ws_bwd_bnapply) /* wsgemm.hip: weight-stationary streaming forms */                   \
  X(wgrad_rect) X(wgrad_square) X(wgrad3) X(wgrad_grouped) X(wgrad_grouped_w3)                                                       \
  X(bn_bwd_apply) X(bn_bwd_apply_capped)                                                        \
  X(knn2) X(knn2_pair) X(knn2_raw) X(knn_rank) X(knn_sel) X(knn_strips) X(knn_big)                                                  \
  X(mr_fwd_lds) X(mr_fwd_grid) X(mr_fwd_key) X(mr_bwd_sorted)                                                                  \
  X(ffn_fused)            /* eval-mode FFN in one launch (ffn_fused.hip) */                     \
  X(mrconv_fused)         /* eval-mode max-relative aggregation + grouped conv in one launch (mrconv_fused.hip) */

enum NsidCounterKey {
#define NSID_CNT_ENUM(name) NSID_C_##name,
  NSID_COUNTER_TABLE(NSID_CNT_ENUM)
#undef NSID_CNT_ENUM
  NSID_C_COUNT
};
extern long g_nsid_counter[NSID_C_COUNT];
static inline void nsid_count(NsidCounterKey k) { ++g_nsid_counter[k]; }

// ---- grouped weight gradients (gemm.hip wgrad_grouped_kernel, wgrad.hip wgrad3_grouped_kernel): the table of problems one launch
// serves, carried in the kernel arguments
constexpr int WGG_MAXP = 28;           // 8 + 116 + 28 x 128 bytes of explicit arguments + 256 hidden: under the 4 KB of a kernel-argument segment
struct WgProb {                 // 128 bytes
  const void* A[2];             // dout of the two row segments (views); [1] unused when seg_splits == nsplit
  const void* B[2];             // x
  const float* bsc[2];          // producer affine of x per segment (or null)
  const float* bsh[2];
  float* C;                     // dw [groups][I][J]
  int lda, ldb, I, J, R;        // R = rows of ONE segment
  int groups, rchunk;           // rows per split
  int seg_splits, nsplit;       // splits of segment 0, splits in all
  int tiles, nwg;               // output tiles per group; workgroups of this problem (the next problem starts at a multiple of 8)
  float slope;
  int pad_period, pad_c1;       // Downsample problems (x read as the zero-padded 3-tap view, see nsid_downsample3_bwd_weight): output nodes per clip, C
};
struct WgGroupArgs {
  int n, pad;
  int wg0[WGG_MAXP + 1];
  WgProb prob[WGG_MAXP];
};
static_assert(sizeof(WgProb) == 128 && sizeof(WgGroupArgs) + 256 <= 4096, "the problem table travels in the kernel arguments");


// workgroup item w of a grouped launch -> problem pi, split (inside its row segment seg), output tile bid, group g; false: padding.
// The tiles of one split read the same rows of both operands, so they go to ONE XCD (item w runs on XCD w % 8 and every problem
// starts at a multiple of 8).
__device__ __forceinline__ bool wgg_decode(const WgGroupArgs& ga, const int w, int& pi, int& split, int& bid, int& g, int& seg) {
  pi = 0;
  for (int i = 1; i < ga.n; ++i) pi += (w >= ga.wg0[i]) ? 1 : 0;          // uniform: scalar loads and compares
  pi = __builtin_amdgcn_readfirstlane(pi);
  const WgProb& q = ga.prob[pi];
  const int l = w - ga.wg0[pi];
  if (l >= q.nwg) return false;
  const int S = q.nsplit;
  int rest;
  if (S < 8 && (8 % S) == 0) {
    const int x = l & 7, per = 8 / S;
    split = x % S;
    rest = (l >> 3) * per + x / S;
  } else {
    split = l % S;
    rest = l / S;
  }
  if (rest >= q.tiles * q.groups) return false;
  g = rest / q.tiles;
  bid = rest % q.tiles;
  seg = split >= q.seg_splits ? 1 : 0;
  split -= seg * q.seg_splits;
  return true;
}
// wgrad.hip: the 8-wave 128x128-tile form of a grouped launch (Nout % 128 == 0, K % 128 == 0, whole 128-row chunks)
int nsid_wgrad3_grouped_launch(const WgGroupArgs& ga, int grid, bool affine, hipStream_t stream);

// wgrad.hip: 128x128-tile form of the bf16 weight-gradient GEMM; returns 1 when the shape is outside its preconditions
int nsid_wgrad2_launch(const void* dout, int ldd, const void* x, int ldx, float* dw, int M, int Nout, int K, int groups,
                       const float* in_scale, const float* in_shift, float slope, hipStream_t stream);

// gemm256.hip: forward GEMM on 256x256 tiles with LDS-DMA staging (bf16 activations and weights, one group); returns 1 when the
// shape is outside its preconditions
int nsid_gemm256_fwd_launch(const void* x, int ldx, const void* w, const float* bias, const void* addend, int ldadd, void* out,
                            int ldo, int M, int Nout, int K, bool relu_out, float* stat, long stat_plane, long stat_ld,
                            hipStream_t stream);
// wsgemm.hip: weight-stationary streaming forward GEMM of the small-K training layers; returns 1 outside its shapes
int nsid_ws_fwd_launch(const void* x, int ldx, const void* w, const float* bias, void* out, int ldo, int M, int Nout, int K,
                       int groups, const float* in_scale, const float* in_shift, float in_slope, float* stat, long stat_plane,
                       long stat_ld, hipStream_t stream);
int nsid_ws_bwd_data_launch(const void* dout, int ldd, const void* w, const void* addend, int ldadd, void* din, int ldi, int M, int Nout,
                            int K, int groups, const void* bn_r, long bn_ldr, const float* bn_scale, const float* bn_shift,
                            const float* bn_mean, const float* bn_invstd, float bn_slope, float* bn_partial, long bn_plane, long bn_ld,
                            const void* abn_r, const float* abn_coef, long abn_plane, float abn_slope, void* abn_dr, long abn_lddr,
                            hipStream_t stream);
// ffn256_fused.hip: eval-mode FFN of the C = 256 stage in one launch; returns 1 outside C = 256, H = 1024, M % 256 == 0
int nsid_ffn256_fused_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int M, int C,
                             int H, hipStream_t stream);
extern void* g_gemm_trace_host;        // gemm.hip: the buffer installed by nsid_debug_gemm_trace (nullptr = none)

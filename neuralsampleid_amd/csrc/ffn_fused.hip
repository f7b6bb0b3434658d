// Eval-mode FFN in ONE launch (forward-only fingerprint extraction, BASELINE config 5):
//     out = x + W2 relu(W1 x + b1) + b2          (encoder/graph_encoder.py:82-89 with both BatchNorms folded into the convs)
// for the C = 64 / 128 stages, where the un-fused pair is HBM-bound: the hidden tensor (M x 4C, bf16) is 4/6 of the FFN's traffic —
// 268 MB written and 268 MB read per launch pair at a 2 048-clip micro-batch — and here it never leaves the CU.
//
// A workgroup (4 waves) owns TM = 128 rows. The hidden dimension is walked in chunks of HC = 128: GEMM 1 (K = C) produces
// relu(W1[chunk] x^T + b1) as D[hidden][row] — the MFMA operands are swapped so that a lane holds 4 CONSECUTIVE hidden values of one
// row and packs them into one 8-byte LDS store in the A-operand layout of GEMM 2 (K = 128), which accumulates D[channel][row] over the
// chunks in registers. LDS images: x / W1 chunk / W2 chunk as R-major rows with a 32-byte pad (fragment reads by ds_read_b128 are
// conflict-free: 16-byte slot (s*row + chunk) mod 16 with s = 2 mod 8 is a bijection over the lane groups the hardware serves together);
// the hidden chunk as [row][128] with the 16-byte slot XOR-swizzled by (row & 15) (reads conflict-free, the 8-byte stores from the
// accumulator layout 2-way). Weights come through L2 (64 KB / 256 KB per 128 rows at C = 64 / 128).
#include "nsid_common.h"

namespace {

struct FfnArgs {
  const __bf16* x; const __bf16* w1; const float* b1; const __bf16* w2; const float* b2; __bf16* out;
  int M, H;
};

constexpr int FF_TM = 128, FF_HC = 128;

// a [ROWS][COLS] bf16 block (global row stride ld elements) on its way into an LDS image with row stride STRIDE bytes, 16 bytes per
// lane: load() only issues the global loads (the next chunk's weights are in flight under the current chunk's MFMAs), store() writes LDS
template <int ROWS, int COLS, int STRIDE, int NT>
struct FfStage {
  static constexpr int CPR = COLS / 8, N = ROWS * CPR, V = N / NT;
  static_assert(N % NT == 0, "whole chunks per thread");
  f32x4 v[V];
  __device__ __forceinline__ void load(const __bf16* __restrict__ g, long ld) {
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int idx = threadIdx.x + NT * q;
      v[q] = *reinterpret_cast<const f32x4*>(g + (long)(idx / CPR) * ld + (idx % CPR) * 8);
    }
  }
  __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int idx = threadIdx.x + NT * q;
      *reinterpret_cast<f32x4*>(lds + (idx / CPR) * STRIDE + (idx % CPR) * 16) = v[q];
    }
  }
};

// NW waves: 2 across the hidden chunk (GEMM 1) / the output channels (GEMM 2), NW / 2 across the 128 rows
template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void ffn_fused_kernel(const FfnArgs p) {
  constexpr int NT = 64 * NW, WI = NW / 2, TI = FF_TM / WI / 16;      // row groups, 16-row tiles per wave
  constexpr int SX = C * 2 + 32;             // row stride of the x and W1-chunk images (K = C)
  constexpr int SW2 = FF_HC * 2 + 32;        // row stride of the W2-chunk image (K = 128)
  constexpr int SH = FF_HC * 2;              // hidden chunk rows: 16 slots of 16 bytes, swizzled
  constexpr int SO = C * 2 + 16;             // output staging rows (reuses the W1-chunk image: free after the last chunk)
  constexpr int TC = C / 32;                 // 16-channel tiles per wave in GEMM 2 (2 waves across the channels)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xa = smem;                           // [TM][SX]
  char* w1c = xa + FF_TM * SX;               // [HC][SX]
  char* w2c = w1c + FF_HC * SX;              // [C][SW2]
  char* hb = w2c + C * SW2;                  // [TM][SH]
  char* ob = w1c;                            // [TM][SO] output staging
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, rq = lane >> 4;
  const int wj = wave / WI, wi = wave % WI;
  const int ib = (FF_TM / WI) * wi;          // first row of this wave's row group
  const long row0 = (long)blockIdx.x * FF_TM;

  FfStage<FF_TM, C, SX, NT> sx;
  FfStage<FF_HC, C, SX, NT> s1;
  FfStage<C, FF_HC, SW2, NT> s2;
  sx.load(p.x + row0 * C, C);
  s1.load(p.w1, C);
  s2.load(p.w2, p.H);
  sx.store(xa);
  s1.store(w1c);
  s2.store(w2c);
  f32x4 acc2[TC][TI];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TI; ++b) acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int h0 = 0; h0 < p.H; h0 += FF_HC) {
    const bool more = h0 + FF_HC < p.H;      // uniform
    if (more) {                              // the next chunk's weights: in flight under this chunk's MFMAs
      s1.load(p.w1 + (long)(h0 + FF_HC) * C, C);
      s2.load(p.w2 + h0 + FF_HC, p.H);
    }
    // ---- GEMM 1: D[j = hidden][i = row] = sum_c W1[j][c] x[i][c]   (A = W1 chunk rows, B = x rows)
    f32x4 acc1[4][TI];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < TI; ++b) acc1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
      bf16x8 fa[4], fb[TI];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(w1c + (64 * wj + 16 * a + lr) * SX + (4 * ks + rq) * 16);
#pragma unroll
      for (int b = 0; b < TI; ++b) fb[b] = *reinterpret_cast<const bf16x8*>(xa + (ib + 16 * b + lr) * SX + (4 * ks + rq) * 16);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) acc1[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc1[a][b], 0, 0, 0);
    }
    // relu(acc + b1) -> bf16, 4 consecutive hidden values of one row per lane: k = 64 wj + 16 a + 4 rq + e, row = ib + 16 b + lr
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 bj = *reinterpret_cast<const f32x4*>(p.b1 + h0 + 64 * wj + 16 * a + 4 * rq);
      const int slot = 8 * wj + 2 * a + (rq >> 1);
#pragma unroll
      for (int b = 0; b < TI; ++b) {
        const int i = ib + 16 * b + lr;
        const f32x4 v = acc1[a][b] + bj;
        bf16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (__bf16)fmaxf(v[e], 0.f);
        *reinterpret_cast<bf16x4*>(hb + i * SH + ((slot ^ (i & 15)) << 4) + (rq & 1) * 8) = h;
      }
    }
    __syncthreads();                         // hidden chunk complete; every wave is done reading the W1 image
    if (more) s1.store(w1c);
    // ---- GEMM 2: D[c][i] += sum_k W2[c][h0 + k] h[i][k]   (A = W2 chunk rows, B = hidden rows)
#pragma unroll
    for (int ks = 0; ks < FF_HC / 32; ++ks) {
      bf16x8 fa[TC], fb[TI];
#pragma unroll
      for (int a = 0; a < TC; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(w2c + ((C / 2) * wj + 16 * a + lr) * SW2 + (4 * ks + rq) * 16);
#pragma unroll
      for (int b = 0; b < TI; ++b) {
        const int i = ib + 16 * b + lr;
        fb[b] = *reinterpret_cast<const bf16x8*>(hb + i * SH + (((4 * ks + rq) ^ (i & 15)) << 4));
      }
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc2[a][b], 0, 0, 0);
    }
    __syncthreads();                         // every wave is done with the hidden buffer and the W2 image
    if (more) s2.store(w2c);                 // (read after the next chunk's first barrier)
  }
  // ---- epilogue: out = acc2 + b2 + x, staged through LDS so that every lane stores 16 contiguous bytes of an output row
#pragma unroll
  for (int a = 0; a < TC; ++a) {
    const int c = (C / 2) * wj + 16 * a + 4 * rq;
    const f32x4 bc = *reinterpret_cast<const f32x4*>(p.b2 + c);
#pragma unroll
    for (int b = 0; b < TI; ++b) {
      const int i = ib + 16 * b + lr;
      const bf16x4 xr = *reinterpret_cast<const bf16x4*>(xa + i * SX + c * 2);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc2[a][b][e] + bc[e] + (float)xr[e]);
      *reinterpret_cast<bf16x4*>(ob + i * SO + c * 2) = o;
    }
  }
  __syncthreads();
  constexpr int CPR = C / 8;
#pragma unroll
  for (int q = 0; q < FF_TM * CPR / NT; ++q) {
    const int idx = threadIdx.x + NT * q;
    const f32x4 v = *reinterpret_cast<const f32x4*>(ob + (idx / CPR) * SO + (idx % CPR) * 16);
    *reinterpret_cast<f32x4*>(p.out + (row0 + idx / CPR) * C + (idx % CPR) * 8) = v;
  }
}

template <int C, int NW>
int ffn_launch(const FfnArgs& p, hipStream_t s) {
  constexpr size_t bytes = (size_t)FF_TM * (C * 2 + 32) + (size_t)FF_HC * (C * 2 + 32) + (size_t)C * (FF_HC * 2 + 32) +
                           (size_t)FF_TM * FF_HC * 2;
  static_assert(bytes <= 160 * 1024 && FF_TM * (C * 2 + 16) <= FF_HC * (C * 2 + 32), "LDS budget");
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel<C, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  NSID_LAUNCH((ffn_fused_kernel<C, NW>), dim3(p.M / FF_TM), dim3(64 * NW), bytes, s, p);
  return nsid_launch_status();
}

}  // namespace

// Returns 1 (nothing launched) for shapes outside the fused form: C in {64, 128}, H = 4C, M % 128 == 0, contiguous bf16 rows.
extern "C" int nsid_ffn_fused_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int M,
                                  int C, int H, void* stream) {
  NSID_REQUIRE(x && w1 && b1 && w2 && b2 && out && M > 0);
  // the x tile in registers (ffn256_fused.hip): C = 256 always (tuning key ffn256), C = 128 / 64 where tuning key ffn_regs has bit 1 / bit 0
  // set and the rows make whole 256- / 512-row tiles
  const long regs = nsid_tune(NSID_T_ffn_regs);
  const bool rx = nsid_tune(NSID_T_ffn256) != 0 && H == 4 * C &&
                  ((C == 256 && M % 256 == 0) || (C == 128 && (regs & 2) != 0 && M % 256 == 0) || (C == 64 && (regs & 1) != 0 && M % 512 == 0));
  if (rx) {
    NSID_REQUIRE(nsid_aligned16(x) && nsid_aligned16(w1) && nsid_aligned16(w2) && nsid_aligned16(out) && nsid_aligned16(b1) &&
                 nsid_aligned16(b2));
    const int rc = nsid_ffn256_fused_launch(x, w1, b1, w2, b2, out, M, C, H, static_cast<hipStream_t>(stream));
    if (rc == NSID_OK) nsid_count(NSID_C_ffn_fused);
    return rc;
  }
  if (!(C == 64 || C == 128) || H != 4 * C || M % FF_TM != 0) return 1;
  NSID_REQUIRE(nsid_aligned16(x) && nsid_aligned16(w1) && nsid_aligned16(w2) && nsid_aligned16(out) && nsid_aligned16(b1) &&
               nsid_aligned16(b2));
  FfnArgs p{static_cast<const __bf16*>(x), static_cast<const __bf16*>(w1), b1, static_cast<const __bf16*>(w2), b2,
            static_cast<__bf16*>(out), M, H};
  nsid_count(NSID_C_ffn_fused);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nsid_tune(NSID_T_ffn_waves) == 4) return C == 64 ? ffn_launch<64, 4>(p, s) : ffn_launch<128, 4>(p, s);
  return C == 64 ? ffn_launch<64, 8>(p, s) : ffn_launch<128, 8>(p, s);
}

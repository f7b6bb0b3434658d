// NOT PART OF THE PRODUCT LIBRARY (round 5 experiment, measured slower inside the two-stream step: docs/experiments.md, "Round 5").
// To time it again: copy to csrc/, add it to build.py SOURCES, declare nsid_wgs_launch + the tuning keys wgrad_stream / wgs_wgs /
// wgs_min_rows / wgs_min_wgs / wgs_slots and the counter wgrad_stream in nsid_common.h, and call nsid_wgs_launch at the top of
// nsid_linear_bwd_weight (rc == 1 falls through to the staged forms). tools/variants/wgrad_stream_test.py is its parity test (62 cases).
//
// Streaming weight gradient of the bf16 path (gfx950): dW[g][Nout][K] += dY[M][g*Nout ..]^T f(X[M][g*K ..]), fp32 accumulation.
//
// Why another body (gemm.hip's 128x64 / 64x64 tiles and wgrad.hip's 8-wave form stage global -> VGPR -> ds_write): a weight gradient
// contracts over ROWS, so both MFMA operands are transposed reads of row-major tiles and every byte crosses the LDS twice; the
// register-staged loops run one or two stages ahead and sit at 40-55 GB/s per CU on the L2 -> LDS path with a third of the workgroup's
// time in its tile-sized burst of fp32 atomics. This form
//   * stages with LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) through a ring of NSLOT 64-row stages
//     (dY tile [64][128] + X tile [64][128] bf16 = 32 KB per slot), NSLOT - 1 stages in flight, ONE workgroup barrier per stage;
//   * keeps both tiles in their global row-major layout; the 64-byte blocks of a 256-byte row are XOR-swizzled by (row & 3) on the
//     SOURCE address (LDS-DMA writes lane-linearly), which makes ds_read_b64_tr_b16 — the hardware transpose that hands a lane 4
//     consecutive rows of its column — conflict-free (4 rows x 64 B of a 32-lane service group fall on 4 distinct bank quarters);
//   * 8 waves = 2 row teams x 4 quadrants (64 x 64 each, 2 x 2 MFMA 32x32x16 tiles, 64 accumulator registers): the teams take the two
//     32-row halves of every stage and their partial tiles meet in LDS at the end, so that a workgroup covers twice the rows per byte
//     of atomics;
//   * a producer BatchNorm + activation of X is applied on the B FRAGMENT: after the transposing read a lane holds 8 rows of ONE
//     channel, so scale / shift are per-lane scalars (packed fp32 math, one rounding to bf16 as the staged forms do).
// Preconditions (host-checked; nsid_wgs_launch returns 1 otherwise): bf16 operands, Nout % 128 == 0, K % 128 == 0 (per group),
// M % (64 * splits) == 0, 16-byte aligned rows.
#include <algorithm>
#include <cstdlib>
#include "nsid_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef bf16x4 __attribute__((address_space(3)))* lds_bf16x4_ptr;

struct WgsArgs {
  const __bf16* dy; long ldd;
  const __bf16* x; long ldx;
  float* dw;
  int Nout, K;                   // per group
  int tiles_k;                   // K / 128
  int rchunk;                    // rows per split (multiple of 64)
  const float* x_scale; const float* x_shift; float x_slope;
};

constexpr int WGS_SLOT = 32768;          // dY image [64 rows][256 B], then X image [64 rows][256 B]

template <int N>
__device__ __forceinline__ void wgs_wait_vm() {
  static_assert(N >= 0 && N < 64, "6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (0xF << 8));
}

// One LDS-DMA in the saddr form (gemm256.hip): 64 lanes x 16 B from (uniform base) + (lane offset) to LDS bytes [dst, dst + 1 KB).
// Not counted by the compiler's vmcnt bookkeeping: every wait of the main loop is explicit.
__device__ __forceinline__ void wgs_glds16(const char* sbase, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

template <int NSLOT, bool AFF>
__global__ __launch_bounds__(512, 2) void wgs_kernel(const WgsArgs p) {
  static_assert(NSLOT >= 3 && NSLOT <= 4, "ring depth");
  __shared__ __attribute__((aligned(1024))) char lds[NSLOT * WGS_SLOT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, quad = wave & 3, qn = quad >> 1, qk = quad & 1;
  const int split = blockIdx.x, tile = blockIdx.y, g = blockIdx.z;
  const int tn = tile / p.tiles_k, tk = tile - tn * p.tiles_k;
  const long r0 = (long)split * p.rchunk;
  const int nst = p.rchunk >> 6;

  // ---- LDS-DMA source: waves 0-3 stage dY rows 16 w .. 16 w + 15 of a stage, waves 4-7 the same rows of X: 4 pieces of 4 rows each
  const bool stage_x = wave >= 4;
  const long ld = stage_x ? p.ldx : p.ldd;
  const int drl = lane >> 4, dpc = lane & 15;
  const int dlc = (((dpc >> 2) ^ drl) << 2) | (dpc & 3);                  // the logical 16-byte chunk that lives at physical chunk dpc
  const unsigned voff = (unsigned)((drl * (int)ld + dlc * 8) * 2);
  const char* const src0 = reinterpret_cast<const char*>(stage_x ? p.x + (r0 + 16 * (wave & 3)) * p.ldx + (long)g * p.K + tk * 128
                                                                 : p.dy + (r0 + 16 * (wave & 3)) * p.ldd + (long)g * p.Nout + tn * 128);
  const long piece_b = ld * 8, stage_b = ld * 128;                        // 4 rows / 64 rows further, in bytes
  const unsigned dst0 = (stage_x ? 16384u : 0u) + (unsigned)(wave & 3) * 4096u;
  auto issue = [&](int st) {
    const char* s = src0 + (long)st * stage_b;
    const unsigned d = (unsigned)(st % NSLOT) * WGS_SLOT + dst0;
#pragma unroll
    for (int i = 0; i < 4; ++i) wgs_glds16(s + i * piece_b, voff, d + i * 1024);
  };

  // ---- fragment addressing (wsgemm.hip ws_bwd_kernel): 16-lane group G = (column half G & 1, row half h = G >> 1); lane 4 qq + pp of a
  // group supplies (row 8 h + qq [+ 4], columns 16 (G & 1) + 4 pp ..) and receives column 16 (G & 1) + (lane & 15) of those 4 rows
  const int G = lane >> 4, h = lane >> 5, qq = (lane >> 2) & 3, pp = lane & 3;
  const unsigned fbase = (unsigned)((32 * team + 8 * h + qq) * 256 + 32 * (G & 1) + 8 * pp);
  unsigned offy[2], offx[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    offy[a] = fbase + (unsigned)(((2 * qn + a) ^ qq) << 6);
    offx[a] = 16384u + fbase + (unsigned)(((2 * qk + a) ^ qq) << 6);
  }
  float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
  if constexpr (AFF) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const long ch = (long)g * p.K + tk * 128 + 64 * qk + 32 * b + (lane & 31);
      sc[b] = p.x_scale[ch];
      sh[b] = p.x_shift[ch];
    }
  }
  const float slope = p.x_slope;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto frag = [&](const char* img, unsigned off) -> bf16x8 {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + off + 4 * 256));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto affine = [&](bf16x8 v, float s_, float t_) -> bf16x8 {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {               // slope in [0, 1]: max(v, v * slope) == (v < 0 ? v * slope : v)
      const f32x2 u = f32x2{s_, s_} * f32x2{(float)v[e], (float)v[e + 1]} + f32x2{t_, t_};
      const f32x2 w = u * slope;
      o[e] = (__bf16)fmaxf(u[0], w[0]);
      o[e + 1] = (__bf16)fmaxf(u[1], w[1]);
    }
    return o;
  };
  auto compute = [&](int st) {
    const char* img = lds + (st % NSLOT) * WGS_SLOT;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fy[2], fx[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fy[a] = frag(img + s * 16 * 256, offy[a]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        fx[b] = frag(img + s * 16 * 256, offx[b]);
        if constexpr (AFF) fx[b] = affine(fx[b], sc[b], sh[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[a], fx[b], acc[a][b], 0, 0, 0);
    }
  };

  // ---- main loop. Invariant at the top of iteration st: stages st .. st + NSLOT - 2 are in flight or landed, slot (st - 1) % NSLOT
  // was read in iteration st - 1. vmcnt counts a wave's own LDS-DMA in issue order (4 per stage); the barrier then says that every wave's
  // pieces of stage st have landed AND that every wave has its fragments of stage st - 1 in registers (an MFMA waits for its operands), so
  // the slot of stage st - 1 takes stage st + NSLOT - 1.
  if constexpr (AFF) wgs_wait_vm<0>();             // the per-lane constants (compiler-counted loads) are not behind the explicit waits
#pragma unroll
  for (int st = 0; st < NSLOT - 1; ++st)
    if (st < nst) issue(st);
  for (int st = 0; st < nst; ++st) {
    const int rem = nst - st;                      // stages not yet consumed, this one included
    if (rem >= NSLOT - 1) wgs_wait_vm<4 * (NSLOT - 2)>();
    else if (NSLOT == 4 && rem == 2) wgs_wait_vm<4>();
    else wgs_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (st + NSLOT - 1 < nst) issue(st + NSLOT - 1);
    compute(st);
  }
  __builtin_amdgcn_s_barrier();                    // the ring is free: it becomes the exchange buffer of the two teams

  // ---- epilogue. C/D layout of the 32x32 tile: column (= k) lane & 31, row (= n) (reg & 3) + 8 (reg >> 2) + 4 h.
  // Team 0 finishes the tiles a = 0, team 1 the tiles a = 1: each parks the half it does not finish ([quad][b][reg][lane] floats,
  // lane-linear: conflict-free), adds the other team's, and issues its share of the atomics.
  float* const park = reinterpret_cast<float*>(lds) + team * 8192 * 2;       // 2 x 32 KB regions
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) park[((quad * 2 + b) * 16 + r) * 64 + lane] = team == 0 ? acc[1][b][r] : acc[0][b][r];
  __syncthreads();
  const float* const take = reinterpret_cast<const float*>(lds) + (1 - team) * 8192 * 2;
  const int a = team;
  float* const wrow = p.dw + (long)g * p.Nout * p.K + (long)(tn * 128 + 64 * qn + 32 * a + 4 * h) * p.K + tk * 128 + 64 * qk + (lane & 31);
  float out[2][16];                                 // every exchange read first, then the atomics back to back
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[b][r] = (team == 0 ? acc[0][b][r] : acc[1][b][r]) + take[((quad * 2 + b) * 16 + r) * 64 + lane];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(wrow + (long)((r & 3) + 8 * (r >> 2)) * p.K + 32 * b, out[b][r]);
}

}  // namespace

// returns 0 / NSID_ELAUNCH, or 1 when the shape is outside this form's preconditions (the caller falls back to gemm.hip / wgrad.hip)
int nsid_wgs_launch(const void* dout, int ldd, const void* x, int ldx, float* dw, int M, int Nout, int K, int groups,
                    const float* in_scale, const float* in_shift, float slope, hipStream_t stream) {
  if (Nout % 128 != 0 || K % 128 != 0 || M % 64 != 0 || ldd % 8 != 0 || ldx % 8 != 0) return 1;
  const long tiles = (long)(Nout / 128) * (K / 128) * groups;
  const long target = nsid_tune(NSID_T_wgs_wgs);
  const long min_rows = std::max<long>(64, nsid_tune(NSID_T_wgs_min_rows));
  long S = std::max<long>(1, target / tiles);
  while (S > 1 && (M % (64 * S) != 0 || M / S < min_rows)) --S;
  if (tiles * S < nsid_tune(NSID_T_wgs_min_wgs)) return 1;
  WgsArgs p{};
  p.dy = static_cast<const __bf16*>(dout); p.ldd = ldd;
  p.x = static_cast<const __bf16*>(x); p.ldx = ldx;
  p.dw = dw;
  p.Nout = Nout; p.K = K;
  p.tiles_k = K / 128;
  p.rchunk = (int)(M / S);
  p.x_scale = in_scale; p.x_shift = in_shift; p.x_slope = slope;
  const dim3 grid((unsigned)S, (unsigned)((Nout / 128) * (K / 128)), (unsigned)groups), block(512);
  const bool aff = in_scale != nullptr;
  if (nsid_tune(NSID_T_wgs_slots) == 3) {
    if (aff) NSID_LAUNCH((wgs_kernel<3, true>), grid, block, 0, stream, p);
    else NSID_LAUNCH((wgs_kernel<3, false>), grid, block, 0, stream, p);
  } else {
    if (aff) NSID_LAUNCH((wgs_kernel<4, true>), grid, block, 0, stream, p);
    else NSID_LAUNCH((wgs_kernel<4, false>), grid, block, 0, stream, p);
  }
  nsid_count(NSID_C_wgrad_stream);
  return nsid_launch_status();
}

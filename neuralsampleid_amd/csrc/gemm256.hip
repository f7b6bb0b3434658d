// Forward row GEMM on 256x256 tiles with LDS-DMA operand staging (gfx950): out[M][N] = act(X[M][K] W[N][K]^T + bias) (+ residual),
// bf16 operands in HBM, fp32 accumulation, bf16 output, optional BatchNorm partial statistics of the fp32 result.
//
// Why a second GEMM body (gemm.hip keeps every other shape): the 128x128-tile kernels stage operands global -> VGPR -> LDS and
// are bound by the bytes a CU can pull through its L2 -> LDS path (~40-70 GB/s per CU). The vendor library's plain
// bf16 product runs the FFN shapes 25-35 % faster than gemm.hip on a 256x256 macro-tile because a stage then moves HALF the
// operand bytes per flop (tools/gemm_bench.py --cold --blas). This kernel takes that tile and stages with
// global_load_lds_dwordx4 (no staging registers, no ds_write pass, no vmcnt -> ds_write dependency):
//   * 512 threads = 8 waves as 2 (rows) x 4 (columns); wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 tiles = 128 accumulator registers;
//   * the reduction advances through a ring of two 64-deep stages = four 32 KB slots (A 256x32 + B 256x32 bf16 each). A stage is
//     fetched by EIGHT LDS-DMA per thread issued together (the two 64-byte halves of every 128-byte operand line back to back),
//     consumed as two 32-deep halves and refilled from the barrier in the middle of the stage: there every wave holds the last
//     fragments of stage t in registers, so stage t's slots take stage t+2, and stage t+1 (issued one stage earlier) must have
//     landed (vmcnt(0): nothing younger is in flight). ONE workgroup barrier per 64-deep stage;
//   * LDS image: [16 rows][32 k] sub-blocks of 1 KB, each written by ONE wave-instruction (LDS-DMA writes lane-linearly:
//     64 lanes x 16 B); the 16-byte chunk index is XOR-ed with 2*(row >> 3 & 1) on the SOURCE address and on the fragment
//     read, which makes every ds_read_b128 lane group {8 rows at chunk c, the other 8 at chunk c+1} hit 16 distinct 16-byte bank
//     slots (MI355X_MICROARCH.md, LDS table: conflict-free, 4 cycles per wave-read);
//   * fragments of the next 32-deep half are read under the MFMAs of the current one: the four B fragments into a second register
//     set, each A fragment IN PLACE right behind the four MFMAs that consume it (64 fragment registers; 96 spilled);
//   * the MFMA operands are swapped (weights first): a lane then holds FOUR CONSECUTIVE COLUMNS of one output row per tile, so the
//     epilogue packs them to bf16 in registers, transposes 16 rows at a time through a private 2.3 KB LDS buffer with ds_write_b64
//     (a quarter of the LDS instructions of a fp32 transpose) and writes whole 128-byte lines;
//   * PERSISTENT: a workgroup walks the tiles b, b + grid, ...; when the main loop of a tile ends the ring is free, so the first two
//     stages of the NEXT tile are issued BEFORE the epilogue (which only touches its own LDS region): the 2-4 us a tile waits for
//     its first operands from HBM and the 1-2 us of epilogue overlap instead of adding up. vmcnt is counted across the boundary.
// Preconditions (host-checked, nsid_gemm256_fwd_launch returns 1 otherwise): M % 256 == 0, N % 256 == 0, K % 128 == 0.
#include <cstdlib>
#include "nsid_common.h"

namespace {

struct G256Args {
  const __bf16* A; long lda;
  const __bf16* B; long ldb;
  __bf16* C; long ldc;
  int M, N, K;
  const float* bias;
  int relu_out;                   // out = max(acc + bias, 0)
  const __bf16* addend; long ldadd;
  float* stat; long stat_plane; long stat_ld;
  unsigned long long* trace;      // nsid_debug_gemm_trace buffer or nullptr: per TILE {start, end of main loop, end, where | prologue << 40}
};

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

constexpr int SLOT = 32768;         // bytes per ring slot: A image [16 row-blocks][1 KB], then B image [16 row-blocks][1 KB]
constexpr int RING = 4 * SLOT;
constexpr int TROW = 144;           // bytes per row of a wave's bf16 transpose buffer: 64 columns + 16 B (rows stay 16-byte aligned)
constexpr int TBUF = 16 * TROW;     // one 16-row batch per wave
constexpr int EPI = 8 * TBUF + 2 * 2 * 256 * 4;      // transpose buffers + parked statistics [2 sums][2 wave-rows][256 columns]
static_assert(RING + EPI <= 160 * 1024, "one workgroup per CU: the ring and the epilogue region together fit the 160 KB of LDS");

// s_waitcnt with the builtin, not inline asm: the compiler's own wait-count bookkeeping then knows what has been waited for
// (simm16 on gfx9: vmcnt [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8]; the fields not meant are left at their maximum)
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (0xF << 8));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// One LDS-DMA in the saddr form: 64 lanes x 16 B from (uniform 64-bit base) + (32-bit lane offset) to LDS bytes [dst, dst + 1 KB).
// Inline asm because the builtin's address, once the compiler has hoisted base + offset out of the loop, becomes a 64-bit vector
// add per instruction: 16 registers the 256-register budget of this kernel does not have (they spilled INTO the main loop).
// M0 (the LDS destination) is written in the statement that reads it and restored (cdna_hip_programming.md, inline-asm rules).
// The compiler does not count this load in its vmcnt bookkeeping: used only inside the main loop, whose waits are all explicit.
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

enum { G256_PLAIN = 0, G256_RELU = 1, G256_STAT = 2, G256_ADD = 3 };     // epilogue variants

template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm256_fwd_kernel(const G256Args p) {
  __shared__ __attribute__((aligned(1024))) char lds[RING + EPI];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, rq = lane >> 4;

  const int tiles_n = p.N >> 8;
  const int ntiles = (p.M >> 8) * tiles_n;
  const int G = gridDim.x;
  // virtual tile v = round * G + block. Blocks b, b + 8 share an XCD (round-robin dispatch), so with ntiles % 8 == 0 XCD x walks the
  // contiguous run [x * ntiles/8, (x+1) * ntiles/8): the column tiles that re-read one A row-panel meet in one L2.
  auto tile_of = [&](int v) { return ((ntiles & 7) == 0 && (G & 7) == 0) ? (v & 7) * (ntiles >> 3) + (v >> 3) : v; };

  // ---- LDS-DMA source addressing: waves 0-3 stage A (row-blocks 4w .. 4w+3), waves 4-7 stage B. Lane l supplies row l >> 2 of
  // the block and the LOGICAL chunk that lives at physical chunk l & 3. Uniform 64-bit base (SGPRs) + ONE 32-bit lane offset for
  // every LDS-DMA of the kernel (saddr form); the row-block and reduction steps are scalar adds.
  const int grow = lane >> 2, lc = (lane & 3) ^ (((grow >> 3) & 1) << 1);
  const bool stage_b = wave >= 4;
  const long ld_s = stage_b ? p.ldb : p.lda;
  const unsigned voff = (unsigned)(grow * (int)ld_s + lc * 8) * 2u;
  const long rb_bytes = ld_s * 32;                        // 16 rows further, in bytes
  const int dst0 = (stage_b ? 16384 : 0) + (wave & 3) * 4096;
  const char* ubase = nullptr;
  auto set_tile = [&](int t, int& i0, int& j0) {
    const int ti = t / tiles_n, tj = t % tiles_n;
    i0 = ti << 8;
    j0 = tj << 8;
    ubase = reinterpret_cast<const char*>(stage_b ? p.B + (long)(j0 + (wave & 3) * 64) * p.ldb
                                                  : p.A + (long)(i0 + (wave & 3) * 64) * p.lda);
  };
  // prologue form (compiler-tracked builtin: the epilogue it overlaps with holds compiler-counted loads); stage t -> slots 2 pair, 2 pair + 1
  auto issue2 = [&](int t, int pair) {
    const char* s2 = ubase + (long)t * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        __builtin_amdgcn_global_load_lds((glb_vptr)(s2 + i * rb_bytes + kb * 64 + voff),
                                         (lds_vptr)(lds + (2 * pair + kb) * SLOT + dst0 + i * 1024), 16, 0, 0);
  };
  const unsigned lds0 = (unsigned)(size_t)(lds_vptr)lds;     // LDS byte address of the ring
  // main-loop form: ONE of the eight LDS-DMA of stage t (q = 2 i + kb: row-block i, 64-byte half kb)
  auto issue1 = [&](int t, int pair, int q) {
    const int i = q >> 1, kb = q & 1;
    glds16(ubase + (long)t * 128 + i * rb_bytes + kb * 64, voff, lds0 + (2 * pair + kb) * SLOT + dst0 + i * 1024);
  };

  // ---- fragment addressing
  const int lo = (lr * 64 + rq * 16) ^ (((lr >> 3) & 1) << 5);
  // four address registers in all: the 16-bit offset field of ds_read reaches slots 0-1 from the first base, slots 2-3 from the second
  const char* fa_base[2] = {lds + wr * 8192 + lo, lds + 2 * SLOT + wr * 8192 + lo};
  const char* fb_base[2] = {lds + 16384 + wc * 4096 + lo, lds + 2 * SLOT + 16384 + wc * 4096 + lo};

  f32x4 acc[4][8];                  // [column tile b][row tile a]: lane (lr, rq) reg r = out[16 a + lr][16 b + 4 rq + r]
  // A fragments: ONE set, refilled in place: fragment a of the next 32-deep half is read right behind the four MFMAs that consume
  // fragment a of the current one (96 fragment registers -> 64). B fragments: two sets.
  bf16x8 fa[8], fb[2][4];
  auto read_a = [&](int a, int slot) {
    fa[a] = *reinterpret_cast<const bf16x8*>(fa_base[slot >> 1] + (slot & 1) * SLOT + a * 1024);
  };
  auto read_b = [&](int set, int slot) {
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[set][b] = *reinterpret_cast<const bf16x8*>(fb_base[slot >> 1] + (slot & 1) * SLOT + b * 1024);
  };

  const int J = p.K >> 5;                                 // 32-deep sub-tiles; a multiple of 4 (K % 128 == 0)
  char* const epi = lds + RING;
  char* const tb = epi + wave * TBUF;                     // this wave's transpose buffer
  float* const red = reinterpret_cast<float*>(epi + 8 * TBUF);
  const int orow = lane >> 3, oq = (lane & 7) * 8;        // bf16 read-back: 8 rows x 8 chunks of 8 columns per pass

  int v = blockIdx.x;
  if (v >= ntiles) return;
  int i0, j0;
  set_tile(tile_of(v), i0, j0);
  unsigned long long t_start = 0, t_first = 0, t_loop = 0;
  if (p.trace) t_start = __builtin_amdgcn_s_memrealtime();
  issue2(0, 0);
  issue2(1, 1);
  bool first = true;
  for (;;) {
    // stage 0 of this tile landed? In flight behind it: stage 1 (8) and, from the second tile on, the 16 stores of the previous
    // epilogue (VMEM operations retire in issue order; the residual loads of that epilogue were consumed, hence retired)
    if (first) wait_vm<8>();
    else wait_vm<8 + (MODE == G256_ADD ? 16 : 16)>();     // 16 stores per thread in either store path
    __builtin_amdgcn_s_barrier();
    if (p.trace) t_first = __builtin_amdgcn_s_memrealtime();
    read_b(0, 0);
#pragma unroll
    for (int a = 0; a < 8; ++a) read_a(a, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[b][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one 32-deep half-step U (slot U): [ODD: barrier] -> B fragments of the next half-step -> 8 x {4 MFMAs of row tile a, A fragment a
    // of the next half-step, [one LDS-DMA of stage t+2]}
#define NSID_G256_STEP(U, ODD, ISSUE, READ)                                                \
  do {                                                                                    \
    if (ODD) {                                                                            \
      wait_lgkm0();                                                                       \
      wait_vm<0>();                                                                       \
      __builtin_amdgcn_s_barrier();                                                       \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    if (READ) read_b(((U) + 1) & 1, ((U) + 1) & 3);                                       \
    _Pragma("unroll") for (int a_ = 0; a_ < 8; ++a_) {                                    \
      _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                    \
        acc[b_][a_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[(U) & 1][b_], fa[a_], acc[b_][a_], 0, 0, 0); \
      if (READ) read_a(a_, ((U) + 1) & 3);                                                \
      if (ISSUE) issue1(((hb + (U)) >> 1) + 2, (U) >> 1, a_);                             \
    }                                                                                     \
    if (READ) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                          \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                  \
      if (READ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
    for (int hb = 0; hb < J - 4; hb += 4) {
      NSID_G256_STEP(0, false, false, true);
      NSID_G256_STEP(1, true, true, true);
      NSID_G256_STEP(2, false, false, true);
      NSID_G256_STEP(3, true, true, true);
    }
    {
      constexpr int hb = 0;          // (only the issue branch, compiled out here, reads it)
      NSID_G256_STEP(0, false, false, true);
      NSID_G256_STEP(1, true, false, true);
      NSID_G256_STEP(2, false, false, true);
      NSID_G256_STEP(3, true, false, false);
    }
#undef NSID_G256_STEP
    // every wave passed the barrier of the last half-step after its last fragment read: the ring is free
    if (p.trace) t_loop = __builtin_amdgcn_s_memrealtime();

    const int ci0 = i0, cj0 = j0, cv = v;        // the tile whose results sit in the accumulators
    // ---------------- epilogue of tile (ci0, cj0). Lane (lr, rq), reg r of acc[b][a]: row 16 a + lr, column 16 b + 4 rq + r.
    // MODE is a template parameter: each variant is straight-line code with its own register budget (as one kernel with run-time
    // flags the residual rows spilled to scratch in EVERY variant and the epilogue took 9.5 us instead of 2.5).
    if (p.bias != nullptr) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const f32x4 bj = *reinterpret_cast<const f32x4*>(p.bias + cj0 + wc * 64 + 16 * b + 4 * rq);
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[b][a] += bj;
      }
    }
    // (statistics variant: thread / lane indices re-derived from the hardware lane count at their use: kept live from the kernel entry
    // across the main loop they spilled into scratch at the 256-register budget)
    const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if constexpr (MODE == G256_STAT) {
      // column sums over the wave's 128 rows: in-lane over the 8 row tiles, then over the 16 lanes lr that share a column
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          s4 += acc[b][a];
          q4 += acc[b][a] * acc[b][a];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = s4[r], q = q4[r];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
          }
          if ((elane & 15) == 0) {
            const int c = wc * 64 + 16 * b + 4 * (elane >> 4) + r;
            red[(0 * 2 + wr) * 256 + c] = s;
            red[(1 * 2 + wr) * 256 + c] = q;
          }
        }
      }
    }
    // ADD: the residual rows of this tile, in the layout of the fp32 read-back below (row = lane >> 2 of a 16-row batch, 8 columns
    // at (lane & 3) * 8 of a 32-column half), requested BEFORE the next tile's LDS-DMA: VMEM data returns in issue order, so issued
    // behind them these loads would wait for that tile's first operands
    f32x4 pre[MODE == G256_ADD ? 16 : 1];
    const long arow0 = (long)(ci0 + wr * 128 + (lane >> 2));
    const int acol = cj0 + wc * 64 + (lane & 3) * 8;
    if constexpr (MODE == G256_ADD) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
          pre[2 * a + bp] = *reinterpret_cast<const f32x4*>(p.addend + (arow0 + 16 * a) * p.ldadd + acol + 32 * bp);
    }
    v += G;
    const bool more = v < ntiles;                // uniform
    if (more) {
      set_tile(tile_of(v), i0, j0);
      issue2(0, 0);
      issue2(1, 1);
    }
    if constexpr (MODE != G256_ADD) {
      // packed to bf16 in registers (the store's rounding), 16 rows at a time through the wave's buffer: ds_write_b64 of four columns,
      // ds_read_b128 of eight, one 16-byte store per lane = whole 128-byte lines per wave-store
      const long crow0 = (long)(ci0 + wr * 128 + orow);
      const int ccol = cj0 + wc * 64 + oq;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          f32x4 x = acc[b][a];
          if constexpr (MODE == G256_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
          }
          *reinterpret_cast<bf16x4*>(tb + lr * TROW + (16 * b + 4 * rq) * 2) = __builtin_convertvector(x, bf16x4);
        }
        // a wave reads back only its own buffer, and one wave's LDS operations execute in issue order: no barrier
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
          *reinterpret_cast<bf16x8*>(p.C + (crow0 + 16 * a + 8 * ps) * p.ldc + ccol) =
              *reinterpret_cast<const bf16x8*>(tb + (ps * 8 + orow) * TROW + oq * 2);
      }
    } else {
      // residual: fp32 through the buffer (ds_write_b128 of four columns, 32 columns at a time), added in fp32, ONE rounding
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) {
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2)
            *reinterpret_cast<f32x4*>(tb + lr * TROW + (16 * b2 + 4 * rq) * 4) = acc[2 * bp + b2][a];
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb + (lane >> 2) * TROW + (lane & 3) * 32);
          const f32x4 t1 = *reinterpret_cast<const f32x4*>(tb + (lane >> 2) * TROW + (lane & 3) * 32 + 16);
          const bf16x8 ad = __builtin_bit_cast(bf16x8, pre[2 * a + bp]);
          float t[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            t[e] = t0[e] + (float)ad[e];
            t[4 + e] = t1[e] + (float)ad[4 + e];
          }
          Chunk<__bf16>::store(p.C + (arow0 + 16 * a) * p.ldc + acol + 32 * bp, t);
        }
    }
    if constexpr (MODE == G256_STAT) {
      __syncthreads();                         // the parked sums of all eight waves (also drains every LDS-DMA: statistics = training,
      const int tid = wave * 64 + elane;       //  one tile per workgroup there)
      if (tid < 256) {
        const long col = cj0 + tid;
        const int cti = ci0 >> 8;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {       // one row of partial sums per 128-row statistics tile = per wave-row
          p.stat[((long)cti * 2 + t2) * p.stat_ld + col] = red[(0 * 2 + t2) * 256 + tid];
          p.stat[p.stat_plane + ((long)cti * 2 + t2) * p.stat_ld + col] = red[(1 * 2 + t2) * 256 + tid];
        }
      }
      __syncthreads();                         // red is reused by the next tile
      wait_vm<0>();                            // the statistics stores are not in the count of the wait at the top
    }
    if (p.trace && wave == 0 && elane == 0) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
      p.trace[4 * (long)cv + 0] = t_start;
      p.trace[4 * (long)cv + 1] = t_loop;
      p.trace[4 * (long)cv + 2] = t_end;
      p.trace[4 * (long)cv + 3] = ((t_first - t_start) << 40) | ((unsigned long long)(xcc & 0xF) << 32) | hw;
      t_start = t_end;
    }
    if (!more) break;
    first = false;
  }
}

}  // namespace

// returns NSID_OK / NSID_ELAUNCH, or 1 when the shape is outside the kernel's preconditions (the caller then takes gemm.hip)
__attribute__((visibility("hidden")))
int nsid_gemm256_fwd_launch(const void* x, int ldx, const void* w, const float* bias, const void* addend, int ldadd, void* out,
                            int ldo, int M, int Nout, int K, bool relu_out, float* stat, long stat_plane, long stat_ld,
                            hipStream_t stream) {
  if (M % 256 != 0 || Nout % 256 != 0 || K % 128 != 0 || ldx % 8 != 0 || ldo % 8 != 0 || (addend && ldadd % 8 != 0)) return 1;
  if (!nsid_aligned16(x) || !nsid_aligned16(w) || !nsid_aligned16(out) || (addend && !nsid_aligned16(addend))) return 1;
  if (bias && !nsid_aligned16(bias)) return 1;
  G256Args p{};
  p.A = static_cast<const __bf16*>(x); p.lda = ldx;
  p.B = static_cast<const __bf16*>(w); p.ldb = K;
  p.C = static_cast<__bf16*>(out); p.ldc = ldo;
  p.M = M; p.N = Nout; p.K = K;
  p.bias = bias;
  p.relu_out = relu_out ? 1 : 0;
  p.addend = static_cast<const __bf16*>(addend); p.ldadd = ldadd;
  p.stat = stat; p.stat_plane = stat_plane; p.stat_ld = stat_ld;
  p.trace = static_cast<unsigned long long*>(g_gemm_trace_host);
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, n = 0;               // an attribute query, not hipGetDeviceProperties: legal whatever the stream is doing (capture)
    n_cu = (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? n : 256;
    if (n_cu <= 0) n_cu = 256;
  }
  const int ntiles = (M / 256) * (Nout / 256);
  const int wgs = nsid_tune(NSID_T_g256_grid) > 0 ? (int)nsid_tune(NSID_T_g256_grid) : n_cu;
  const dim3 grid(ntiles < wgs ? ntiles : wgs);       // one workgroup per CU (155 KB of LDS), persistent over the tiles
  if (addend && (stat || relu_out)) return 1;
  if (stat && relu_out) return 1;
  if (addend) NSID_LAUNCH(gemm256_fwd_kernel<G256_ADD>, grid, dim3(512), 0, stream, p);
  else if (stat) NSID_LAUNCH(gemm256_fwd_kernel<G256_STAT>, grid, dim3(512), 0, stream, p);
  else if (relu_out) NSID_LAUNCH(gemm256_fwd_kernel<G256_RELU>, grid, dim3(512), 0, stream, p);
  else NSID_LAUNCH(gemm256_fwd_kernel<G256_PLAIN>, grid, dim3(512), 0, stream, p);
  return nsid_launch_status();
}

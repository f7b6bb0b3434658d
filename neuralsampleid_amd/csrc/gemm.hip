// Row GEMMs of the Grapher/FFN stack on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, one
// rounding per product — bitwise a k-ordered fmaf chain, so results are tile-shape independent).
//
// One kernel template, three operand layouts (A = left operand, B = right operand, R = reduction index):
//   forward      out = f(X) W^T   A = X  [i][R] R-contiguous   B = W    [j][R] R-contiguous
//   backward-data dX = dY W       A = dY [i][R] R-contiguous   B = W    [R][j] j-contiguous
//   backward-wgt  dW = dY^T f(X)  A = dY [R][i] i-contiguous   B = X    [R][j] j-contiguous  (R = rows, split + atomics)
// Fusions: BatchNorm-apply + activation of the PRODUCER layer on operand load ("f"), bias / ELU / residual-gradient
// addend in the epilogue, and the per-row-tile column sums / sums of squares that training-mode BatchNorm of THIS
// layer needs, so a conv+BN+act layer costs one read of its input and one write of its raw output.
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; wave tile (BM/2)x(BN/2) of 16x16 MFMA tiles; BK = 16 per stage,
// register-staged double buffer in LDS. Each lane fetches 4 consecutive reduction elements of its fragment row with
// one ds_read_b128 and feeds them to 4 MFMAs (sub-step s uses element s); A and B use the same permutation
// r = 4*(lane>>4) + s inside the 16-deep chunk, so the sum is unchanged.
#include <algorithm>
#include <cstdlib>
#include "nsid_common.h"

namespace {

inline float act_slope(int act) { return act == NSID_ACT_RELU ? 0.f : (act == NSID_ACT_LEAKY ? 0.2f : 1.f); }

struct GemmArgs {
  const void* A; long lda; long a_goff;   // group offset in elements (operand storage: fp32 or bf16, see ST)
  const void* B; long ldb; long b_goff;
  void* C; long ldc; long c_goff;
  int I, J, R;
  // operand-load activation as a negative-side slope (1 = none, 0 = ReLU, 0.2 = LeakyReLU): one branch-free select
  const float* a_scale; const float* a_shift; float a_slope; long a_aff_goff;   // per-R affine on A (forward)
  const float* b_scale; const float* b_shift; float b_slope; long b_aff_goff;   // per-j affine on B (backward-wgt)
  const float* bias; long bias_goff;                                        // per-j
  const void* addend; long ldadd;                                           // same indexing and storage as C
  float* stat; long stat_plane;    // stat[0*plane + tile*stat_ld + col], stat[1*plane + ...]
  long stat_ld;
  int rsplit;     // number of R-splits (grid.y); >1 => atomic epilogue
  int rchunk;     // R elements per split (multiple of 16)
  int atomic_out;
  int split_major;  // grid = (splits, tiles): the tiles of one split share an XCD (blocks b, b+8 share an L2)
  // backward-data only: the output is dL/dy of a BatchNorm(+activation) layer whose raw input r is given; the epilogue
  // also emits that layer's backward column sums (what nsid_bn_bwd_reduce would compute from the stored output)
  const void* bn_r; long bn_ldr;
  const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd; float bn_slope;
  float* bn_partial; long bn_plane; long bn_ld;
  // Zero-padded strided views (Downsample as a 3-tap stride-2 GEMM over x itself, no im2col): elements of the padded
  // operand whose row index r satisfies r % pad_period == pad_phase and whose column lies in [pad_c0, pad_c1) read as 0
  // and are never fetched (they lie in the neighbouring clip or outside the tensor). "row" = i (R-major A) or the
  // reduction index (i/j-major B); pad_safe = an element offset that is always inside the operand.
  int pad_period, pad_phase, pad_c0, pad_c1; long pad_safe;
  int relu_out;   // forward, non-atomic store path: out = max(acc + bias, 0) (eval mode: the consumer then needs no activation on load)
  // ABN (backward-data, bf16 storage, full tiles): the left operand is the BatchNorm BACKWARD of the layer in front, evaluated on
  // the operand load from TWO tensors of the same layout — A = dy (gradient w.r.t. that layer's activated output) and abn_r = r (its
  // raw conv output): dr = sc*g + P*r + Q with g = dy * act'(sc*r + sh); abn_coef[4][abn_plane] = {sc, sh, P, Q} per channel
  // (nsid_bn_bwd_finalize_fused). The workgroups of column tile 0 also write dr (abn_dr, row stride abn_lddr) for the weight
  // gradient: the separate bn_bwd_apply pass (read dy, read r, write dr) disappears.
  const void* abn_r; const float* abn_coef; long abn_plane; float abn_slope; void* abn_dr; long abn_lddr;
};

// Precision H = false: fp32 operands, v_mfma_f32_16x16x4_f32, BK = 16 (exact fp32 — the parity path).
// Precision H = true : operands rounded to bf16 (RNE, v_cvt_pk_bf16_f32) as they are written to LDS, fp32 storage in
//                      HBM and fp32 accumulation, v_mfma_f32_16x16x32_bf16 (16x the fp32 matrix rate), BK = 32.
template <bool H> struct Prec { static constexpr int BK = H ? 32 : 16; static constexpr int ESZ = H ? 2 : 4; };

// KS: 32-deep MFMA sub-steps per stage of the bf16 path (BK = 32*KS). KS = 2 doubles the bytes a workgroup keeps in flight
// and halves the barriers per reduction element: the GEMMs with <= 2 workgroups per CU (the C = 256 / 512 stages: 256-512
// tiles) were fetching at 16 GB/s per CU against the 60-70 GB/s the L2 -> LDS path delivers (round-2 shape table).
template <int ROWS, bool RMAJOR, bool H, bool SRC16 = false, int KS = 1, int NT = 256>
struct TileGeom {
  static_assert(KS == 1 || H, "deeper stages exist on the bf16 path only");
  static constexpr int BK = Prec<H>::BK * KS, ESZ = Prec<H>::ESZ;
  // LDS images whose fragment READS are bank-conflict-free (MI355X_MICROARCH.md, LDS: banking is per instruction, over fixed lane
  // groups). The staging WRITES are not: a ds_write_b128 is served 8 lanes at a time, and 8 consecutive 16-byte chunks of 96-byte rows
  // (32-deep stages) fold 2-way onto the 32 write banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.22-0.31 on the KS = 1 kernels,
  // 0.04-0.08 with 160-byte rows (KS = 2) (profiles/r02d_final/pmc_sq_summary.txt). Those replays stay under the 13 cycles the write
  // instruction spends moving its operands to the LDS anyway (guide: a store conflict costs time only once the array cycles exceed
  // the instruction's own) -- measured in round 4 (tools/lds_stage_write_bench.hip, profiles/r04d/lds_stage_write_bench.jsonl): this
  // stage's two ds_write_b128 per thread + barrier cost 241.8 cycles per stage at pitch 96 (2-way), at pitch 80 (conflict-free writes)
  // and at pitch 128 alike (256.0 each with three workgroups per CU), while the fragment reads + MFMAs cost 343 cycles at pitch 96
  // against 424 at pitch 80 (404 / 516 with three workgroups) -- so the row pitch is chosen for the reads:
  //  RMAJOR: lds[row][BK elements + 32 B pad] (row = i or j, R contiguous): 96-byte rows in both precisions (160-byte rows
  //          with KS = 2: 16-byte block (10*lr + rq) mod 16 is a bijection for the same lane split). A fragment
  //          read is one ds_read_b128 per lane at (row lr, 16-byte chunk rq); the hardware serves lanes {0-3,12-15,20-27}
  //          etc. together, i.e. all 16 rows with chunk rq on half of them and rq+1 on the other half: 16-byte block
  //          (6*lr + rq) mod 16 is a bijection for that split (80-byte rows, 5*lr + rq, collide 3 ways per group: half of
  //          the LDS cycles of the forward kernels were conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.43).
  //  else  : lds[BK][ROWS elements + 32 B pad] (R = row, i/j contiguous), bf16: every block of 8 reduction rows is shifted
  //          by a further 128 B. ds_read_b64_tr_b16 serves 32 lanes together = reduction rows {0-3} and {8-11} (4 x 8 B
  //          each): row starts r*STRIDE + (r>>3)*128 put the 8 rows on 8 disjoint 32-byte bank ranges for ROWS = 64 and 128
  //          (with a plain stride, rows r and r+8 — or r and r+1 — always share banks).
  static constexpr int SHIFT8 = (!RMAJOR && H) ? 128 : 0;                  // extra bytes per block of 8 reduction rows
  static constexpr int STRIDE = RMAJOR ? BK * ESZ + 32 : ROWS * ESZ + (H ? 32 : 16);   // bytes
  static constexpr int BYTES = (RMAJOR ? ROWS : BK) * STRIDE + (BK / 8) * SHIFT8;
  // SRC16: the operand is stored as bf16 in HBM -> one 16-byte load carries 8 elements instead of 4
  static constexpr int EPC = SRC16 ? 8 : 4;             // elements per 16-byte source chunk
  static constexpr int SSZ = SRC16 ? 2 : 4;             // source element size
  static constexpr int VEC = ROWS * BK / EPC / NT;      // 16-byte chunks per thread per stage (NT threads per workgroup)
  static_assert(ROWS * BK / EPC % NT == 0, "a stage is a whole number of chunks per thread");
  static constexpr int CPR = BK / EPC;                  // chunks per row of an R-major tile
  static constexpr int CPC = ROWS / EPC;                // chunks per (reduction) row of an i/j-major tile
  static_assert(!SRC16 || H, "bf16 storage implies the bf16 MFMA path");
};

// Staging is split in two so that the global-load latency hides under the MFMA block of the current stage
// (issue early / write late): stage_load only ISSUES the 16-byte loads; stage_store, which runs after the MFMAs,
// applies affine + activation, zero-fills out-of-range chunks, rounds to bf16 in the H path, and writes LDS.
// A bf16-stored operand without affine is copied chunk-for-chunk (its HBM image IS the LDS image).
template <int ROWS, bool RMAJOR, bool H, bool SRC16, int KS = 1, int NT = 256>
struct StageRegs {
  static constexpr int VEC = TileGeom<ROWS, RMAJOR, H, SRC16, KS, NT>::VEC;
  f32x4 v[VEC];      // raw 16-byte chunks (4 fp32 or 8 bf16)
  bool ok[VEC];
};

// FULL: the tile and every stage lie inside the operand (host-checked), so there is no predication at all — the bounds
// logic (compare, select, zero-fill per chunk) is a third of the instructions of a K = 256 tile otherwise.
template <int ROWS, bool RMAJOR, bool H, bool SRC16, bool FULL, int KS = 1, bool PAD = false, int NT = 256>
__device__ __forceinline__ void stage_load(StageRegs<ROWS, RMAJOR, H, SRC16, KS, NT>& s, const char* __restrict__ base, long ld,
                                           int row0, int nrows, int r0, int rend, const GemmArgs* pad = nullptr) {
  using G = TileGeom<ROWS, RMAJOR, H, SRC16, KS, NT>;
  static_assert(!PAD || !FULL, "padded views take the predicated path");
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < G::VEC; ++q) {
    const int idx = t + NT * q;
    long off;
    if (RMAJOR) {
      const int gi = row0 + idx / G::CPR, gr = r0 + (idx % G::CPR) * G::EPC;
      s.ok[q] = FULL || (gi < nrows && gr < rend);    // extents are multiples of the chunk: all-in or all-out
      if (PAD) s.ok[q] = s.ok[q] && !(gi % pad->pad_period == pad->pad_phase && gr >= pad->pad_c0 && gr < pad->pad_c1);
      off = s.ok[q] ? (long)gi * ld + gr : (PAD ? pad->pad_safe : 0);   // skipped chunks read a safe element, zeroed later
    } else {
      const int gr = r0 + idx / G::CPC, gc = row0 + (idx % G::CPC) * G::EPC;
      s.ok[q] = FULL || (gr < rend && gc < nrows);
      if (PAD) s.ok[q] = s.ok[q] && !(gr % pad->pad_period == pad->pad_phase && gc >= pad->pad_c0 && gc < pad->pad_c1);
      off = s.ok[q] ? (long)gr * ld + gc : (PAD ? pad->pad_safe : 0);
    }
    s.v[q] = *reinterpret_cast<const f32x4*>(base + off * G::SSZ);
  }
}

// reduction-indexed affine (producer BatchNorm) of an R-major operand for the stage that starts at r0. VMEM completes in
// order, so these small loads must be issued BEFORE the operand loads of a later stage: issued after them, the first
// use would wait for the whole prefetch (vmcnt is positional) and collapse the pipeline.
template <int ROWS, bool H, bool SRC16, int KS = 1, int NT = 256>
__device__ __forceinline__ void affine_prefetch(f32x4* sc, f32x4* sh, int r0, int rend, const float* scale,
                                                const float* shift) {
  using G = TileGeom<ROWS, true, H, SRC16, KS, NT>;
#pragma unroll
  for (int q = 0; q < G::VEC; ++q) {
    const int gr = r0 + ((threadIdx.x + NT * q) % G::CPR) * G::EPC;
    const int ga = gr < rend ? gr : 0;
#pragma unroll
    for (int e = 0; e < G::EPC; e += 4) {        // raw float4 registers: nothing consumes them before the commit
      sc[(q * G::EPC + e) >> 2] = *reinterpret_cast<const f32x4*>(scale + ga + e);
      sh[(q * G::EPC + e) >> 2] = *reinterpret_cast<const f32x4*>(shift + ga + e);
    }
  }
}

// column-indexed affine of an i/j-major operand is the same for every stage: fetched once per kernel
template <int ROWS, bool H, bool SRC16, int KS = 1, int NT = 256>
__device__ __forceinline__ void colaffine_load(f32x4* sc, f32x4* sh, int row0, int nrows, const float* scale,
                                               const float* shift) {
  using G = TileGeom<ROWS, false, H, SRC16, KS, NT>;
#pragma unroll
  for (int q = 0; q < G::VEC; ++q) {
    const int cv = ((threadIdx.x + NT * q) % G::CPC) * G::EPC;
    const int gc = row0 + cv < nrows ? row0 + cv : 0;
#pragma unroll
    for (int e = 0; e < G::EPC; e += 4) {
      sc[(q * G::EPC + e) >> 2] = *reinterpret_cast<const f32x4*>(scale + gc + e);
      sh[(q * G::EPC + e) >> 2] = *reinterpret_cast<const f32x4*>(shift + gc + e);
    }
  }
}

// ReLU on eight bf16 values in their storage format: as signed 16-bit integers every negative float (sign bit set) is a
// negative integer and every non-negative float a non-negative one, so max(x, 0) on int16 lanes IS ReLU (v_pk_max_i16: one
// instruction per two elements; -0 -> +0; a NaN with the sign bit set becomes 0, one without passes)
__device__ __forceinline__ f32x4 relu_bf16x8(f32x4 raw) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 h = __builtin_bit_cast(s16x8, raw);
  h = __builtin_elementwise_max(h, s16x8{0, 0, 0, 0, 0, 0, 0, 0});
  return __builtin_bit_cast(f32x4, h);
}

template <int ROWS, bool RMAJOR, bool H, bool SRC16, bool RELU16 = false, int KS = 1, int NT = 256>
__device__ __forceinline__ void stage_store(char* lds, const StageRegs<ROWS, RMAJOR, H, SRC16, KS, NT>& s, bool affine,
                                            float slope, const f32x4* csc, const f32x4* csh) {
  using G = TileGeom<ROWS, RMAJOR, H, SRC16, KS, NT>;
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < G::VEC; ++q) {
    const int idx = t + NT * q;
    const int lo = RMAJOR ? (idx / G::CPR) * G::STRIDE + (idx % G::CPR) * G::EPC * G::ESZ
                          : (idx / G::CPC) * G::STRIDE + ((idx / G::CPC) >> 3) * G::SHIFT8 +
                                (idx % G::CPC) * G::EPC * G::ESZ;
    if (SRC16 && !affine) {           // wave-uniform: bf16 in HBM == bf16 in LDS
      f32x4 raw = s.v[q];
      if (!s.ok[q]) raw = f32x4{0.f, 0.f, 0.f, 0.f};
      if (RELU16) raw = relu_bf16x8(raw);
      *reinterpret_cast<f32x4*>(lds + lo) = raw;
      continue;
    }
    float x[G::EPC];
    if (SRC16) {
      const bf16x8 h = __builtin_bit_cast(bf16x8, s.v[q]);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (float)h[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = s.v[q][e];
    }
    if (affine) {
#pragma unroll
      for (int e = 0; e < G::EPC; e += 2) {
        // csc/csh: per-chunk affine registers — column-indexed (fetched once per kernel) for i/j-major tiles,
        // reduction-indexed (prefetched per stage by affine_prefetch, AHEAD of the next operand loads) for R-major ones.
        // Packed fp32 math (v_pk_fma / v_pk_mul); slope in [0, 1]: v < 0 ? v*slope : v == max(v, v*slope), NaN stays NaN
        const f32x4 c4 = csc[(q * G::EPC + e) >> 2], h4 = csh[(q * G::EPC + e) >> 2];
        const f32x2 v = f32x2{c4[e & 3], c4[(e & 3) + 1]} * f32x2{x[e], x[e + 1]} + f32x2{h4[e & 3], h4[(e & 3) + 1]};
        const f32x2 w = v * slope;
        x[e] = fmaxf(v[0], w[0]);
        x[e + 1] = fmaxf(v[1], w[1]);
      }
    }
    if (!s.ok[q]) {
#pragma unroll
      for (int e = 0; e < G::EPC; ++e) x[e] = 0.f;
    }
    if (H) {
#pragma unroll
      for (int e = 0; e < G::EPC; e += 4)
        *reinterpret_cast<bf16x4*>(lds + lo + e * 2) =
            __builtin_convertvector((f32x4{x[e], x[e + 1], x[e + 2], x[e + 3]}), bf16x4);
    } else {
      *reinterpret_cast<f32x4*>(lds + lo) = f32x4{x[0], x[1], x[2], x[3]};
    }
  }
}

// ABN staging of an R-major bf16 tile: dr = sc*g + (P*r + Q), g = z > 0 ? dy : dy*slope, z = sc*r + sh, rounded to bf16 once (the value
// bn_bwd_apply would have stored), written to the LDS image and — for the workgroups of column tile 0 — to the dr side output.
// Every chunk of a thread covers the SAME 8 reduction channels (NT % CPR == 0), so the four coefficient vectors are 8 registers each.
template <int ROWS, int KS, int NT>
__device__ __forceinline__ void stage_store_abn(char* lds, const StageRegs<ROWS, true, true, true, KS, NT>& dy,
                                                const StageRegs<ROWS, true, true, true, KS, NT>& rr, const f32x4* cf, float slope,
                                                char* side, long side_ld) {
  using G = TileGeom<ROWS, true, true, true, KS, NT>;
  static_assert(NT % G::CPR == 0, "a thread keeps its reduction channels over the chunks of a stage");
  const int t = threadIdx.x;
  const bool masked = slope != 1.f;          // uniform: a BatchNorm without an activation behind it needs no mask (g = dy)
#pragma unroll
  for (int q = 0; q < G::VEC; ++q) {
    const int idx = t + NT * q;
    const int row = idx / G::CPR, kc = idx % G::CPR;
    const bf16x8 hd = __builtin_bit_cast(bf16x8, dy.v[q]), hr = __builtin_bit_cast(bf16x8, rr.v[q]);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f32x4 sc4 = cf[0 + (e >> 2)], sh4 = cf[2 + (e >> 2)], p4 = cf[4 + (e >> 2)], q4 = cf[6 + (e >> 2)];
      const int l = e & 3;
      const f32x2 x = {(float)hr[e], (float)hr[e + 1]}, d = {(float)hd[e], (float)hd[e + 1]};
      const f32x2 sc = {sc4[l], sc4[l + 1]};
      f32x2 g = d;
      if (masked) {
        const f32x2 z = sc * x + f32x2{sh4[l], sh4[l + 1]};
        const f32x2 ds = d * slope;
        g = f32x2{z[0] > 0.f ? d[0] : ds[0], z[1] > 0.f ? d[1] : ds[1]};
      }
      const f32x2 v = sc * g + (f32x2{p4[l], p4[l + 1]} * x + f32x2{q4[l], q4[l + 1]});
      o[e] = (__bf16)v[0];
      o[e + 1] = (__bf16)v[1];
    }
    *reinterpret_cast<bf16x8*>(lds + row * G::STRIDE + kc * 16) = o;
    if (side != nullptr) *reinterpret_cast<bf16x8*>(side + ((long)row * side_ld + kc * 8) * 2) = o;
  }
}

// fp32 fragment of one 16-row tile: element s feeds MFMA sub-step s (reduction index 4*(lane>>4)+s of the chunk)
template <int ROWS, bool RMAJOR>
__device__ __forceinline__ f32x4 frag_read_f32(const char* lds, int row, int rq) {
  using G = TileGeom<ROWS, RMAJOR, false>;
  if (RMAJOR) {
    return *reinterpret_cast<const f32x4*>(lds + row * G::STRIDE + 16 * rq);
  } else {
    f32x4 f;
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const float*>(lds + (4 * rq + s) * G::STRIDE + 4 * row);
    return f;
  }
}

// bf16 fragment for v_mfma_f32_16x16x32_bf16: lane (lr, rq) holds 8 consecutive reduction elements 8*rq .. 8*rq+7 of
// tile row lr.  R-major tiles: one ds_read_b128.  i/j-major tiles: two ds_read_b64_tr_b16 — per 16-lane group the
// hardware gathers a 4 (reduction) x 16 (i/j) block and hands every lane its column, i.e. a free transpose; lane
// 4q+p of the group supplies the address of block row q, columns 4p..4p+3 (EXEC is all ones here: no divergence).
template <int ROWS, bool RMAJOR, int KS = 1>
__device__ __forceinline__ bf16x8 frag_read_bf16(const char* lds, int tile_row0, int lr, int rq, int ks = 0) {
  using G = TileGeom<ROWS, RMAJOR, true, false, KS>;
  if (RMAJOR) {
    return *reinterpret_cast<const bf16x8*>(lds + (tile_row0 + lr) * G::STRIDE + 16 * rq + 64 * ks);
  } else {
    typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
    const char* base = lds + (32 * ks + 8 * rq + (lr >> 2)) * G::STRIDE + (4 * ks + rq) * G::SHIFT8 +
                       (tile_row0 + 4 * (lr & 3)) * 2;
    const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(base));
    const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(base + 4 * G::STRIDE));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// a raw 16-byte chunk as floats: 8 bf16 (BF) or 4 fp32
template <bool BF>
__device__ __forceinline__ void chunk_to_float(const f32x4& raw, float* v) {
  if constexpr (BF) {
    const bf16x8 h = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = raw[e];
  }
}

// ST: ACTIVATION tensors are bf16 in HBM. The left operand is always an activation; the right operand is one only in
// the weight-gradient variant (both i/j-major); the output is an activation in the forward and backward-data variants.
// AAFF: the left operand carries a reduction-indexed affine (forward GEMM fed by a raw conv output). A template flag,
// not a runtime one: a runtime branch around the per-stage affine loads makes the compiler's vmcnt bookkeeping
// conservative at the join and drains the prefetch.
// WB: the right operand is a WEIGHT matrix stored as bf16 (the optimiser's shadow copy, optim.py): its HBM image is the
// LDS image — half the L2->LDS bytes of fp32 weights and no convert pass (forward / backward-data variants only).
// Debug timeline (tools/gemm_trace.py): when a buffer is installed with nsid_debug_gemm_trace(), every workgroup records
// {start, end of main loop, end} on the 100 MHz constant clock plus where it ran (HW_ID, XCC_ID) — 4 x u64 per workgroup.
// One scalar load and a uniform branch per workgroup when no buffer is installed.
__device__ unsigned long long* g_gemm_trace = nullptr;

// ARELU: the left operand is a bf16 activation that only needs ReLU on load (an eval-mode BatchNorm folded into the producer's
// weights leaves no affine): applied on the packed bf16 values, no conversion, no per-channel vectors.
// KS / PD: 32-deep sub-steps per stage and register sets (= stages in flight) of the pipelined loop; PD = 0 keeps the round-1
// rule (2 sets, 1 for the full-tile forward kernel). The deep forms (KS = 2 and/or PD = 4) are FULL-tile bf16 variants for the
// shapes that put <= 2 workgroups on a CU, where only bytes in flight per workgroup hide the memory latency.
// NW: waves per workgroup, laid out (NW/2) x 2 over the tile. NW = 8 on 256x128 tiles gives every wave the 64x64 sub-tile
// of the 4-wave 128x128 kernel while a stage moves a quarter fewer operand bytes per flop, at two workgroups = 16 waves
// per CU (the 4-wave 256x128 form needs 201 registers: 8 waves per CU).
// The body is a device function of (arguments, tile index, split index, group): gemm_kernel derives the three from its own grid,
// wgrad_grouped_kernel (below) from a table of many weight-gradient problems served by ONE launch.
template <int BM, int BN, bool A_RMAJOR, bool B_RMAJOR, bool H, bool ST, bool AAFF, bool WB = false, bool FULL = false,
          bool ARELU = false, int KS = 1, int PD = 0, bool EC = false, int PADX = 0, int NW = 4, bool ABN = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int bid, const int split, const int g) {
  unsigned long long* const trace = g_gemm_trace;
  unsigned long long t_start = 0, t_loop = 0;
  if (trace) t_start = __builtin_amdgcn_s_memrealtime();
  constexpr bool SA = ST, SB = (ST && !A_RMAJOR && !B_RMAJOR) || WB, SC = ST && A_RMAJOR;
  static_assert(!WB || (ST && A_RMAJOR), "bf16 weights ride with bf16 activations in the forward/backward-data GEMMs");
  static_assert(!ARELU || (ST && A_RMAJOR && B_RMAJOR && !AAFF && FULL), "ReLU-on-load is a forward, full-tile, bf16 variant");
  static_assert((KS == 1 && PD == 0) || (FULL && H && ST), "the deep pipelines are full-tile bf16-storage variants");
  static_assert(!EC || PD >= 2, "early commit rides with the deep pipelines");
  static_assert(NW == 4 || NW == 8, "4 waves as 2x2 or 8 waves as 4x2");
  static_assert(!ABN || (FULL && H && ST && WB && A_RMAJOR && !B_RMAJOR && !AAFF && !ARELU && !EC && PADX == 0 && NW == 4),
                "the BatchNorm-backward operand load is a full-tile bf16 backward-data variant");
  constexpr int NT = 64 * NW, WROWS = NW / 2;
  using GA = TileGeom<BM, A_RMAJOR, H, SA, KS, NT>;
  using GB = TileGeom<BN, B_RMAJOR, H, SB, KS, NT>;
  constexpr int BK = Prec<H>::BK * KS;
  constexpr int WM = BM / WROWS, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  // BatchNorm partial sums are per NSID_ROW_TILE = 128 rows: a tile of BM rows covers STILES of them, each made of WPT wave-rows
  constexpr int STILES = BM >= 128 ? BM / 128 : 1, WPT = WROWS / STILES;
  constexpr int STAGE = GA::BYTES + GB::BYTES;
  constexpr int RB = NW == 8 ? 16 : 32;                // rows a wave transposes through LDS at a time (epilogue)
  constexpr int OUT_STAGE = NW * RB * (WN + 4) * 4     // epilogue transpose buffers (one per wave x RB rows), bytes
                            + 2 * WROWS * 4 * BN * 4;  // + parked BatchNorm sums [2][wave-rows][4 row groups][BN]
  // EC (early commit): THREE stage buffers, so that stage s+1 can be written to LDS while stage s is still being read: the
  // loop then issues the fragment reads of stage s, commits stage s+1 (its waits and ds_writes run under the read latency)
  // and only then starts the MFMAs — in the two-buffer order (MFMA, then commit, then barrier) a wave's LDS-read latency,
  // its ds_write completion and the barrier were all exposed: ~800 cycles per 32-deep slab against 256 cycles of MFMA.
  constexpr int NBUF = EC ? 3 : 2;
  constexpr int LDS_BYTES = NBUF * STAGE > OUT_STAGE ? NBUF * STAGE : OUT_STAGE;
  __shared__ __attribute__((aligned(16))) char lds_raw[LDS_BYTES];
  float* lds = reinterpret_cast<float*>(lds_raw);

  const int tiles_j = (p.J + BN - 1) / BN;
  const int tiles_i = (p.I + BM - 1) / BM;
  const int ti = bid / tiles_j, tj = bid % tiles_j;
  if (ti >= tiles_i) return;
  const int i0 = ti * BM, j0 = tj * BN;
  const int rbeg = split * p.rchunk;
  const int rend = min(p.R, rbeg + p.rchunk);

  const char* A = reinterpret_cast<const char*>(p.A) + g * p.a_goff * GA::SSZ;
  const char* B = reinterpret_cast<const char*>(p.B) + g * p.b_goff * GB::SSZ;
  const float* a_sc = p.a_scale ? p.a_scale + g * p.a_aff_goff : nullptr;
  const float* a_sh = p.a_shift ? p.a_shift + g * p.a_aff_goff : nullptr;
  const float* b_sc = p.b_scale ? p.b_scale + g * p.b_aff_goff : nullptr;
  const float* b_sh = p.b_shift ? p.b_shift + g * p.b_aff_goff : nullptr;

  // readfirstlane: the wave index is uniform, and only an SGPR tells the compiler so (row/column bases derived from it then
  // stay scalar instead of becoming per-lane 64-bit multiplies)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lr = lane & 15, rq = lane >> 4;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // only these operand/affine pairings exist: forward (A reduction-major, affine on the reduction index) and
  // backward-weight (B column-major, affine on the column index)
  constexpr bool a_aff = AAFF;
  const bool b_aff = !B_RMAJOR && b_sc != nullptr;
  f32x4 bcs[GB::VEC * GB::EPC / 4], bch[GB::VEC * GB::EPC / 4];
  if (b_aff) colaffine_load<BN, H, SB, KS, NT>(bcs, bch, j0, p.J, b_sc, b_sh);
  const int nstage = (rend - rbeg + BK - 1) / BK;

  // Register prefetch depth: fp32 MFMA blocks (64 x 32 cycles) cover one memory round trip, so one stage in flight
  // is enough; the bf16 MFMA block (16 x 16 cycles) is far shorter than the round trip, so TWO stages are kept in
  // flight (two register sets, static indices through the 2x unrolled loop body below).
  // Exception: the full-tile forward kernel with bf16 weights and no operand affine keeps ONE stage in flight: measured
  // 7-10 % faster than the deeper prefetch when introduced and equal to it now (tools/gemm_trace.py: 3 workgroups per CU
  // are resident either way, the loop is bound by the L2 -> LDS rate, ~53 GB/s per CU, not by load latency).
  constexpr int DEPTH = PD > 0 ? PD : ((H && !(FULL && WB && !AAFF && BN == 128 && A_RMAJOR && B_RMAJOR)) ? 2 : 1);
  static_assert(DEPTH == 1 || DEPTH == 2 || DEPTH == 4, "the unrolled loop body needs an even number of register sets");
  StageRegs<BM, A_RMAJOR, H, SA, KS, NT> ra[DEPTH];
  StageRegs<BN, B_RMAJOR, H, SB, KS, NT> rb[DEPTH];
  StageRegs<BM, A_RMAJOR, H, SA, KS, NT> ra2[ABN ? DEPTH : 1];     // ABN: the second source tensor (r) of the left operand
  f32x4 abn_cf[ABN ? 8 : 1];                                        // ABN: {sc, sh, P, Q} x 8 channels of the stage being committed
  const char* A2 = ABN ? reinterpret_cast<const char*>(p.abn_r) + g * p.a_goff * GA::SSZ : nullptr;

  f32x4 acs[GA::VEC * GA::EPC / 4], ach[GA::VEC * GA::EPC / 4];   // reduction-indexed affine of the stage being committed
  // FULL tiles address a stage as (uniform 64-bit base in SGPRs) + (32-bit lane offset fixed for the whole kernel): the
  // loads take the saddr form and cost no vector instructions. Per-chunk 64-bit row*ld products were 14 VALU (5 of them
  // quarter rate) per stage of the weight-gradient loop against 4 MFMAs — the GEMM family is VALU-issue bound at these
  // reduction lengths (tools/asm_profile.py), not LDS- or memory-bound.
  unsigned voa[GA::VEC], vob[GB::VEC];
  if constexpr (FULL) {
#pragma unroll
    for (int q = 0; q < GA::VEC; ++q) {
      const int idx = threadIdx.x + NT * q;
      voa[q] = A_RMAJOR ? (unsigned)(((idx / GA::CPR) * (int)p.lda + (idx % GA::CPR) * GA::EPC) * GA::SSZ)
                        : (unsigned)(((idx / GA::CPC) * (int)p.lda + (idx % GA::CPC) * GA::EPC) * GA::SSZ);
    }
#pragma unroll
    for (int q = 0; q < GB::VEC; ++q) {
      const int idx = threadIdx.x + NT * q;
      vob[q] = B_RMAJOR ? (unsigned)(((idx / GB::CPR) * (int)p.ldb + (idx % GB::CPR) * GB::EPC) * GB::SSZ)
                        : (unsigned)(((idx / GB::CPC) * (int)p.ldb + (idx % GB::CPC) * GB::EPC) * GB::SSZ);
    }
  }
  auto issue = [&](auto& sa, auto& sb, auto& sa2, int st) {
    int r0 = rbeg + st * BK;
    if constexpr (FULL) {
      r0 = min(r0, rend - BK);      // the two prefetches past the last stage re-read it (never computed on)
      const char* pa = A + (A_RMAJOR ? (long)i0 * p.lda + r0 : (long)r0 * p.lda + i0) * GA::SSZ;
      const char* pb = B + (B_RMAJOR ? (long)j0 * p.ldb + r0 : (long)r0 * p.ldb + j0) * GB::SSZ;
#pragma unroll
      for (int q = 0; q < GA::VEC; ++q) {
        sa.v[q] = *reinterpret_cast<const f32x4*>(pa + voa[q]);
        sa.ok[q] = true;
      }
      if constexpr (ABN) {
        const char* pa2 = A2 + ((long)i0 * p.lda + r0) * GA::SSZ;       // same layout and row stride as A (host-checked)
#pragma unroll
        for (int q = 0; q < GA::VEC; ++q) sa2.v[q] = *reinterpret_cast<const f32x4*>(pa2 + voa[q]);
      }
#pragma unroll
      for (int q = 0; q < GB::VEC; ++q) {
        sb.v[q] = *reinterpret_cast<const f32x4*>(pb + vob[q]);
        sb.ok[q] = true;
      }
    } else {
      stage_load<BM, A_RMAJOR, H, SA, FULL, KS, PADX == 1, NT>(sa, A, p.lda, i0, p.I, r0, rend, &p);
      stage_load<BN, B_RMAJOR, H, SB, FULL, KS, PADX == 2, NT>(sb, B, p.ldb, j0, p.J, r0, rend, &p);
    }
  };
  auto aff_fetch = [&](int st) {
    if constexpr (AAFF) affine_prefetch<BM, H, SA, KS, NT>(acs, ach, rbeg + st * BK, rend, a_sc, a_sh);
    if constexpr (ABN) {       // the 8 channels of this thread's chunks in stage st (past the end: the last stage again, unused)
      const int r0 = min(rbeg + st * BK, rend - BK) + (int)(threadIdx.x % GA::CPR) * 8;
      const float* c = p.abn_coef + g * p.a_goff + r0;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        abn_cf[2 * v] = *reinterpret_cast<const f32x4*>(c + v * p.abn_plane);
        abn_cf[2 * v + 1] = *reinterpret_cast<const f32x4*>(c + v * p.abn_plane + 4);
      }
    }
  };
  auto commit = [&](const auto& sa, const auto& sb, const auto& sa2, int st) {
    char* dst = lds_raw + (EC ? st % 3 : (st & 1)) * STAGE;
    if constexpr (ABN) {
      // column tile 0 of every row panel also stores dr (each stage exactly once: phantom stages past the end write LDS only)
      char* side = (tj == 0 && st < nstage && p.abn_dr != nullptr)
                       ? reinterpret_cast<char*>(p.abn_dr) + ((long)i0 * p.abn_lddr + g * p.a_goff + rbeg + st * BK) * 2 : nullptr;
      stage_store_abn<BM, KS, NT>(dst, sa, sa2, abn_cf, p.abn_slope, side, p.abn_lddr);
    } else
    stage_store<BM, A_RMAJOR, H, SA, ARELU, KS, NT>(dst, sa, a_aff, p.a_slope, acs, ach);
    stage_store<BN, B_RMAJOR, H, SB, false, KS, NT>(dst + GA::BYTES, sb, b_aff, p.b_slope, bcs, bch);
  };
  // early-commit form of one stage: the fragment reads of stage st, the global loads of stage st+DEPTH (`loads()`), the commit
  // of stage st+1 (`between()`) and the MFMAs of stage st are ONE scheduling region, and the sched_group_barrier sequence
  // asks for: all fragment reads first, then {2 MFMAs, 1 global load, 1 ds_write} repeated — a wave that issues its 16
  // global loads back to back sits in the memory pipeline's issue queue for hundreds of cycles (a 1 KB wave-load takes the
  // CU's vector memory path >= 16 cycles) and, alone on its SIMD, issues no MFMA meanwhile.
  auto compute_ec = [&](int st, auto&& loads, auto&& between) {
    const char* la = lds_raw + (st % 3) * STAGE;
    const char* lb = la + GA::BYTES;
    bf16x8 fa[KS][TM], fb[KS][TN];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[ks][a] = frag_read_bf16<BM, A_RMAJOR, KS>(la, wm0 + 16 * a, lr, rq, ks);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[ks][b] = frag_read_bf16<BN, B_RMAJOR, KS>(lb, wn0 + 16 * b, lr, rq, ks);
    }
    loads();
    between();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][a], fb[ks][b], acc[a][b], 0, 0, 0);
    constexpr int NMFMA = KS * TM * TN, NLOAD = GA::VEC + GB::VEC;
    constexpr int NDSR = KS * (TM * (A_RMAJOR ? 1 : 2) + TN * (B_RMAJOR ? 1 : 2));
    __builtin_amdgcn_sched_group_barrier(0x100, NDSR, 0);
    constexpr int PER = NMFMA / NLOAD > 0 ? NMFMA / NLOAD : 1;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };
  auto compute = [&](int st) {
    const char* la = lds_raw + (st & 1) * STAGE;
    const char* lb = la + GA::BYTES;
    if (H) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[a] = frag_read_bf16<BM, A_RMAJOR, KS>(la, wm0 + 16 * a, lr, rq, ks);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[b] = frag_read_bf16<BN, B_RMAJOR, KS>(lb, wn0 + 16 * b, lr, rq, ks);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    } else {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = frag_read_f32<BM, A_RMAJOR>(la, wm0 + 16 * a + lr, rq);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = frag_read_f32<BN, B_RMAJOR>(lb, wn0 + 16 * b + lr, rq);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
    }
  };

  if (DEPTH == 1) {
    if (nstage > 0) {
      aff_fetch(0);
      issue(ra[0], rb[0], ra2[ABN ? (0) : 0], 0);
      commit(ra[0], rb[0], ra2[ABN ? (0) : 0], 0);
    }
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
      const bool more = st + 1 < nstage;
      if (more) {                                   // lands under the MFMA block below
        aff_fetch(st + 1);
        issue(ra[0], rb[0], ra2[ABN ? (0) : 0], st + 1);
      }
      compute(st);
      if (more) commit(ra[0], rb[0], ra2[ABN ? (0) : 0], st + 1);
      __syncthreads();
    }
  } else {
    // Two stages in flight, branch-free body. A stage index past the end addresses r0 >= rend: every chunk is out of
    // range, reads a dummy element and is committed as zeros, so phantom stages add exactly 0 to the accumulators and
    // the loop can always run an even number of stages with NO control flow between a load and its wait — the
    // compiler's vmcnt(N) is then exact (a join would force vmcnt(0) and drain the prefetch).
    // Invariant at the top of an iteration: LDS[st&1] holds stage st, register set 1 holds stage st+1 (in flight),
    // register set 0 is free. Plain loads stay in flight across __syncthreads(); the scheduling barriers pin the
    // phase order issue -> MFMA -> commit.
    // DEPTH register sets: at the top of sub-step u of an iteration, LDS[s&1] holds stage s = st+u, set u is free (it held
    // stage s) and the other DEPTH-1 sets hold stages s+1 .. s+DEPTH-1 in flight.
    aff_fetch(0);
    issue(ra[0], rb[0], ra2[ABN ? (0) : 0], 0);
    commit(ra[0], rb[0], ra2[ABN ? (0) : 0], 0);
#pragma unroll
    for (int d = 1; d < DEPTH; ++d) issue(ra[d], rb[d], ra2[ABN ? (d) : 0], d);
    __syncthreads();
    for (int st = 0; st < nstage; st += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        if constexpr (EC) {
          aff_fetch(st + u + 1);
          compute_ec(st + u, [&] { issue(ra[u], rb[u], ra2[ABN ? (u) : 0], st + u + DEPTH); },
                     [&] { commit(ra[(u + 1) % DEPTH], rb[(u + 1) % DEPTH], ra2[ABN ? ((u + 1) % DEPTH) : 0], st + u + 1); });
        } else {
          aff_fetch(st + u + 1);
          issue(ra[u], rb[u], ra2[ABN ? (u) : 0], st + u + DEPTH);
          __builtin_amdgcn_sched_barrier(0);
          compute(st + u);
          __builtin_amdgcn_sched_barrier(0);
          commit(ra[(u + 1) % DEPTH], rb[(u + 1) % DEPTH], ra2[ABN ? ((u + 1) % DEPTH) : 0], st + u + 1);
        }
        __syncthreads();
      }
    }
  }

  if (trace) t_loop = __builtin_amdgcn_s_memrealtime();
  // ---------------- epilogue. C/D layout: col = lane&15, row = 4*(lane>>4) + reg.
  // The reduction loops here are 2-64 stages long, so the epilogue's vector instructions count as much as the loop's
  // (tools/asm_profile.py: 630 of the 900 VALU instructions of a K = 256 forward tile were epilogue). Hence: the bias is
  // folded into the accumulators once behind a uniform branch, the statistics use packed fp32 math and no cross-lane
  // shuffles (all four row groups park their sums in LDS), and every global access is uniform base + fixed lane offset.
  float* C = reinterpret_cast<float*>(p.C) + g * p.c_goff;      // fp32 view: the atomic path always writes fp32
  const float* bias = (p.bias && split == 0) ? p.bias + g * p.bias_goff : nullptr;
  if (bias != nullptr) {             // wave-uniform
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int j = j0 + wn0 + 16 * b + lr;
      const float bj = (FULL || j < p.J) ? bias[j] : 0.f;
#pragma unroll
      for (int a = 0; a < TM; ++a) acc[a][b] += bj;
    }
  }
  if (p.atomic_out) {
    // split-reduction outputs (weight gradients, split-K): one atomic per element
    float* crow = C + (long)(i0 + wm0 + 4 * rq) * p.ldc + (j0 + wn0 + lr);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const bool jok = FULL || j0 + wn0 + 16 * b + lr < p.J;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + wm0 + 16 * a + 4 * rq + r;
          if ((FULL || i < p.I) && jok) atomicAdd(crow + (long)(16 * a + r) * p.ldc + 16 * b, acc[a][b][r]);
        }
    }
  } else {
    // Stores go through LDS so that every lane writes 16 contiguous bytes of an output row (4-byte stores from the
    // accumulator layout are store-issue bound). Each wave transposes its own 64 x WN sub-tile, 32 rows at a time.
    constexpr int OLD = WN + 4;                  // staggers rq groups over the banks, keeps rows 16-B aligned
    float* ost = lds + wave * (RB * OLD);        // one RB x (WN+4) float buffer per wave
    constexpr int OE = SC ? 8 : 4;               // output elements per lane per store (16 bytes either way)
    constexpr int OSZ = SC ? 2 : 4;
    constexpr int Q_PER_ROW = WN / OE, ROWS_PER_PASS = 64 / Q_PER_ROW;
    const int orow = lane / Q_PER_ROW, oq = (lane % Q_PER_ROW) * OE;
    const int jq = j0 + wn0 + oq;
    const bool jqok = FULL || jq < p.J;          // J % OE == 0: a chunk is all-in or all-out
    char* Cb = reinterpret_cast<char*>(p.C) + g * p.c_goff * OSZ;
    const char* Ab = reinterpret_cast<const char*>(p.addend) + g * p.c_goff * OSZ;
    if (p.stat != nullptr) {
      // BatchNorm partial statistics of (acc + bias) from the accumulator registers, parked in LDS behind the transpose
      // buffers BEFORE the store loop: its barrier publishes them and the stores hide the LDS latency
      float* red = lds + NW * RB * OLD;      // [2 sums][wave-rows][4 row groups][BN]
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const bool jok = FULL || j0 + wn0 + 16 * b + lr < p.J;
        f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          f32x4 v = acc[a][b];
          if (!FULL) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (!(i0 + wm0 + 16 * a + 4 * rq + r < p.I && jok)) v[r] = 0.f;
          }
          const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
          s2 += lo; s2 += hi;
          q2 += lo * lo; q2 += hi * hi;
        }
        const int c = wn0 + 16 * b + lr;
        red[((0 * WROWS + (wave >> 1)) * 4 + rq) * BN + c] = s2[0] + s2[1];
        red[((1 * WROWS + (wave >> 1)) * 4 + rq) * BN + c] = q2[0] + q2[1];
      }
    }
    // fused BatchNorm-backward column sums (backward-data, bf16 output): g = dy * act'(scale*r+shift), xhat = (r-mean)*invstd
    constexpr bool CAN_BNRED = A_RMAJOR && !B_RMAJOR && SC;
    const bool bnred = CAN_BNRED && p.bn_r != nullptr;        // wave-uniform
    const bool bn_unit = p.bn_slope == 1.f;                   // no activation between the BatchNorm and this gradient
    float bsc[OE], bsh[OE], bmu[OE], bis[OE];
    f32x2 s0[OE / 2], s1[OE / 2];
    if (CAN_BNRED) {
#pragma unroll
      for (int e = 0; e < OE; ++e) bsc[e] = bsh[e] = bmu[e] = bis[e] = 0.f;
#pragma unroll
      for (int e = 0; e < OE / 2; ++e) { s0[e] = f32x2{0.f, 0.f}; s1[e] = f32x2{0.f, 0.f}; }
      if (bnred && jqok) {
        const long ch = g * p.c_goff + jq;
        load_channels<OE>(p.bn_scale, (int)ch, bsc);
        load_channels<OE>(p.bn_shift, (int)ch, bsh);
        load_channels<OE>(p.bn_mean, (int)ch, bmu);
        load_channels<OE>(p.bn_invstd, (int)ch, bis);
      }
    }
    // lane offsets inside a pass are fixed; a pass starts at a uniform row
    const unsigned lo_c = (unsigned)((orow * (int)p.ldc + oq) * OSZ);
    const unsigned lo_a = (unsigned)((orow * (int)p.ldadd + oq) * OSZ);
    const unsigned lo_r = (unsigned)((orow * (int)p.bn_ldr + oq) * 2);
    // Backward-data only: the residual-gradient addend and the BatchNorm input of every store pass are fetched NOW, all at
    // once. Loaded inside the pass they were two dependent HBM round trips per pass: 8 passes, one workgroup per CU and cold
    // caches made that epilogue 12.6 us behind a 19 us main loop (tools/gemm_trace.py --addend --bn 1 --cold).
    constexpr bool CAN_ADD = A_RMAJOR;      // backward-data (residual gradient) and forward (residual stream, nsid_linear_fwd_res)
    constexpr int PPH = RB / ROWS_PER_PASS, NPASS = (WM / RB) * PPH;
    static_assert(RB % ROWS_PER_PASS == 0 && WM % RB == 0, "a transpose batch is whole store passes");
    f32x4 pre_a[CAN_ADD ? NPASS : 1], pre_r[CAN_BNRED ? NPASS : 1];
    if constexpr (CAN_ADD) {
      if (p.addend) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
          const int ib = i0 + wm0 + RB * (q / PPH) + (q % PPH) * ROWS_PER_PASS;
          const bool ok = (FULL || ib + orow < p.I) && jqok;
          pre_a[q] = *reinterpret_cast<const f32x4*>(Ab + (ok ? ((long)ib * p.ldadd + (j0 + wn0)) * OSZ + lo_a : 0));
        }
      }
    }
    if constexpr (CAN_BNRED) {
      if (bnred) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
          const int ib = i0 + wm0 + RB * (q / PPH) + (q % PPH) * ROWS_PER_PASS;
          const bool ok = (FULL || ib + orow < p.I) && jqok;
          pre_r[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.bn_r) +
                                                     (ok ? ((long)ib * p.bn_ldr + g * p.c_goff + (j0 + wn0)) * 2 + lo_r : 0));
        }
      }
    }
#pragma unroll
    for (int h = 0; h < WM / RB; ++h) {
#pragma unroll
      for (int a2 = 0; a2 < RB / 16; ++a2)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) ost[(16 * a2 + 4 * rq + r) * OLD + 16 * b + lr] = acc[(RB / 16) * h + a2][b][r];
      // no workgroup barrier: a wave reads back only its own transpose buffer, and the LDS operations of one wave execute in
      // issue order (four barriers per tile were pure skew)
#pragma unroll
      for (int pass = 0; pass < RB / ROWS_PER_PASS; ++pass) {
        const int rr = pass * ROWS_PER_PASS + orow;
        const int ib = i0 + wm0 + RB * h + pass * ROWS_PER_PASS;          // uniform first row of the pass
        if ((FULL || ib + orow < p.I) && jqok) {
          float v[OE];
#pragma unroll
          for (int e = 0; e < OE; e += 4) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(ost + rr * OLD + oq + e);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) v[e + k4] = t4[k4];
          }
          if (p.relu_out) {              // uniform
#pragma unroll
            for (int e = 0; e < OE; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if constexpr (CAN_ADD) {
            if (p.addend) {
              float ad[OE];
              chunk_to_float<SC>(pre_a[h * PPH + pass], ad);
#pragma unroll
              for (int e = 0; e < OE; ++e) v[e] += ad[e];
            }
          }
          char* cp = Cb + ((long)ib * p.ldc + (j0 + wn0)) * OSZ + lo_c;
          if (SC) Chunk<__bf16>::store(reinterpret_cast<__bf16*>(cp), v);
          else Chunk<float>::store(reinterpret_cast<float*>(cp), v);
          if (CAN_BNRED) {
            if (bnred) {
              float x[OE];
              chunk_to_float<true>(pre_r[h * PPH + pass], x);
#pragma unroll
              for (int e = 0; e < OE; e += 2) {
                // dy: the value a separate reduce pass would read back (bf16-rounded)
                const f32x2 dy = {(float)(__bf16)v[e], (float)(__bf16)v[e + 1]};
                const f32x2 xx = {x[e], x[e + 1]};
                f32x2 gg = dy;
                if (!bn_unit) {
                  const f32x2 z = f32x2{bsc[e], bsc[e + 1]} * xx + f32x2{bsh[e], bsh[e + 1]};
                  const f32x2 ds = dy * p.bn_slope;
                  gg[0] = z[0] > 0.f ? dy[0] : ds[0];
                  gg[1] = z[1] > 0.f ? dy[1] : ds[1];
                }
                s0[e / 2] += gg;
                s1[e / 2] += gg * ((xx - f32x2{bmu[e], bmu[e + 1]}) * f32x2{bis[e], bis[e + 1]});
              }
            }
          }
        }
      }
    }
    __syncthreads();                // every wave is done with its transpose buffer; publishes the parked statistics
    if (CAN_BNRED) {
      if (bnred) {                  // uniform; the stage / transpose buffers are free (barrier above)
        float* red2 = lds;          // [2][NW waves][ROWS_PER_PASS][WN]
#pragma unroll
        for (int e = 0; e < OE; ++e) {
          red2[((0 * NW + wave) * ROWS_PER_PASS + orow) * WN + oq + e] = s0[e / 2][e & 1];
          red2[((1 * NW + wave) * ROWS_PER_PASS + orow) * WN + oq + e] = s1[e / 2][e & 1];
        }
        __syncthreads();
        if (threadIdx.x < BN) {
          const int c = threadIdx.x, half = c / WN, cw = c % WN;
          const int j = j0 + c;
          if (FULL || j < p.J) {
            const long col = g * p.c_goff + j;
#pragma unroll
            for (int t2 = 0; t2 < STILES; ++t2) {           // one row of partial sums per 128-row statistics tile
              float a0 = 0.f, a1 = 0.f;
#pragma unroll
              for (int wr = WPT * t2; wr < WPT * (t2 + 1); ++wr)
#pragma unroll
                for (int o = 0; o < ROWS_PER_PASS; ++o) {   // unrolled: the LDS reads pipeline instead of one round trip each
                  a0 += red2[((0 * NW + (2 * wr + half)) * ROWS_PER_PASS + o) * WN + cw];
                  a1 += red2[((1 * NW + (2 * wr + half)) * ROWS_PER_PASS + o) * WN + cw];
                }
              p.bn_partial[((long)ti * STILES + t2) * p.bn_ld + col] = a0;
              p.bn_partial[p.bn_plane + ((long)ti * STILES + t2) * p.bn_ld + col] = a1;
            }
          }
        }
      }
    }
  }
  if (p.stat != nullptr) {   // uniform branch: the column sums were parked in LDS before the store loop (see above)
    const float* red = lds + NW * RB * ((BN / 2) + 4);
    if (threadIdx.x < BN) {
      const int j = j0 + threadIdx.x;
      if (FULL || j < p.J) {
        const long col = g * p.c_goff + j;     // forward: c_goff == Nout per group == column offset
#pragma unroll
        for (int t2 = 0; t2 < STILES; ++t2) {  // one row of partial sums per 128-row statistics tile
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int k = 4 * WPT * t2; k < 4 * WPT * (t2 + 1); ++k) {      // WPT wave-rows x 4 row groups, fixed order
            s += red[(0 * WROWS * 4 + k) * BN + threadIdx.x];
            q += red[(1 * WROWS * 4 + k) * BN + threadIdx.x];
          }
          p.stat[((long)ti * STILES + t2) * p.stat_ld + col] = s;
          p.stat[p.stat_plane + ((long)ti * STILES + t2) * p.stat_ld + col] = q;
        }
      }
    }
  }
  if (trace && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long lin = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
    trace[4 * lin + 0] = t_start;
    trace[4 * lin + 1] = t_loop;
    trace[4 * lin + 2] = __builtin_amdgcn_s_memrealtime();
    trace[4 * lin + 3] = ((unsigned long long)xcc << 32) | hw;
  }
}

template <int BM, int BN, bool A_RMAJOR, bool B_RMAJOR, bool H, bool ST, bool AAFF, bool WB = false, bool FULL = false,
          bool ARELU = false, int KS = 1, int PD = 0, bool EC = false, int PADX = 0, int NW = 4, bool ABN = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : ((KS > 1 || PD > 2) ? 1 : ((FULL && WB && BM == 128 && BN == 128 && !AAFF && !ABN) ? 3 : 2)))   // waves per SIMD
void gemm_kernel(const GemmArgs p) {
  int bid = p.split_major ? blockIdx.y : blockIdx.x;
  const int split = p.split_major ? blockIdx.x : blockIdx.y;
  if (!p.split_major) {
    // XCD-aware remap (blocks b and b+8 share an L2): give each XCD a contiguous run of tiles so the
    // column tiles that re-read one A row-panel hit the same L2. Bijective only when the grid divides by 8.
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
  }
  gemm_body<BM, BN, A_RMAJOR, B_RMAJOR, H, ST, AAFF, WB, FULL, ARELU, KS, PD, EC, PADX, NW, ABN>(p, bid, split, blockIdx.z);
}

// ---- many weight-gradient problems in ONE launch (the deferred phase of a training step: nsid_linear_bwd_weight_grouped) ----------
// A weight gradient hangs off the dependent chain — nothing reads it before the optimiser — but as a launch of its own it fills 256 CUs
// only by splitting its row reduction 8-16 ways (fp32 atomics: workgroups x tile bytes) and pays launch latency, first-operand latency
// and a tail once per layer and view. Here the problems of many layers form one grid: a workgroup looks its problem up in a table that
// travels in the kernel arguments (nothing is retained, the launch is capturable), both views of a layer are two row SEGMENTS of one
// problem (a split lies in one segment: the body's loop is untouched), and because the union of the problems fills the chip each
// problem needs only rows / wgg_rows splits: a quarter of the atomic bytes of the per-layer launches.
template <int BM, int BN, bool FULL, bool H = true, bool ST = true, int PADX = 0>
__device__ __forceinline__ void wgrad_grouped_item(const WgGroupArgs& ga, const int w);

// The grid is min(workgroups of all problems, cap): with a cap a workgroup walks the items b, b + grid, b + 2 grid, ... (a static
// schedule: no counter, every workgroup reaches its exit). A capped launch runs BESIDE the two backward chains of a step (one
// workgroup per CU leaves the other view's kernels their LDS and wave slots); the final, uncapped one has the chip to itself.
// H / ST: bf16 MFMA operands / bf16 storage (the projector head keeps fp32 tensors: ST = false; strict fp32 arithmetic: H = false);
// PADX = 2: x is a Downsample input read as the zero-padded 3-tap view.
template <int BM, int BN, bool FULL, bool H = true, bool ST = true, int PADX = 0>
__global__ __launch_bounds__(256, FULL ? 1 : 2) void wgrad_grouped_kernel(const WgGroupArgs ga) {
  const int total = ga.wg0[ga.n];
  for (int w = blockIdx.x; w < total; w += gridDim.x) wgrad_grouped_item<BM, BN, FULL, H, ST, PADX>(ga, w);
}

template <int BM, int BN, bool FULL, bool H, bool ST, int PADX>
__device__ __forceinline__ void wgrad_grouped_item(const WgGroupArgs& ga, const int w) {
  int pi, split, bid, g, seg;
  if (!wgg_decode(ga, w, pi, split, bid, g, seg)) return;
  const WgProb& q = ga.prob[pi];
  GemmArgs p{};
  p.A = q.A[seg]; p.lda = q.lda; p.a_goff = q.I;
  p.B = q.B[seg]; p.ldb = q.ldb; p.b_goff = q.J;
  p.C = q.C; p.ldc = q.J; p.c_goff = (long)q.I * q.J;
  p.I = q.I; p.J = q.J; p.R = q.R;
  p.b_scale = q.bsc[seg]; p.b_shift = q.bsh[seg]; p.b_slope = q.slope; p.b_aff_goff = q.J;
  p.a_slope = 1.f; p.bn_slope = 1.f;
  p.atomic_out = 1;
  p.rchunk = q.rchunk; p.rsplit = q.nsplit;
  if constexpr (PADX == 2) {
    p.pad_period = q.pad_period; p.pad_phase = 0; p.pad_c0 = 0; p.pad_c1 = q.pad_c1; p.pad_safe = q.pad_c1;
  }
  gemm_body<BM, BN, false, false, H, ST, false, false, FULL, false, FULL ? 2 : 1, FULL ? 2 : 0, false, PADX>(p, bid, split, g);
  // (the body's reduction loop ends with a workgroup barrier behind the last fragment reads and its atomic epilogue does not touch
  // LDS: the next item of a capped launch may stage its first operands right away)
}

// ELU of the projector (simclr/simclr.py:26) runs as its own in-place pass: expm1f inlined into the fully unrolled
// GEMM epilogue would bloat every instantiation for a 256 x 4096 tensor.
__global__ void elu_inplace_kernel(float* __restrict__ x, long rows, int cols, long ld) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float* q = x + (i / cols) * ld + (i % cols);
    const float v = *q;
    *q = v > 0.f ? v : expm1f(v);
  }
}

int g_gemm_precision = NSID_GEMM_FP32;     // process-wide (nsid_set_gemm_precision)
// Tuning keys g256_min / g256_train: smallest number of 256x256 tiles for which the forward GEMM takes gemm256.hip (LDS-DMA
// staging); launches WITH a statistics epilogue (training) take it only when g256_train = 1 — the default serves forward-only work,
// so that the arithmetic of a training step does not depend on the batch size.
long g_g256_launches = 0;     // launches that took gemm256.hip (tests check that the kernel under test really ran)

template <int BM, int BN, bool AR, bool BR, int NW = 4>
int launch(GemmArgs p, int groups, hipStream_t s, int act_dtype, bool w_bf16 = false) {
  const int tiles = ((p.I + BM - 1) / BM) * ((p.J + BN - 1) / BN);
  const bool st16 = act_dtype == NSID_BF16;
  const bool half = g_gemm_precision == NSID_GEMM_BF16 || st16;
  const int bk = half ? 32 : 16;
  p.rchunk = (p.rchunk + bk - 1) / bk * bk;            // whole stages per split
  p.rsplit = (p.R + p.rchunk - 1) / p.rchunk;
  // a split re-reads nothing another split reads; tiles of ONE split share both operand row-panels. With the split
  // index fastest (and a multiple of 8 of them) consecutive blocks go to different XCDs and one split stays on one L2.
  p.split_major = p.rsplit > 1 && (p.rsplit % 8) == 0;
  dim3 grid(p.split_major ? p.rsplit : tiles, p.split_major ? tiles : p.rsplit, groups);
  constexpr bool CAN_AFF = AR && BR;                   // only the forward variant takes a reduction-indexed affine
  const bool aff = CAN_AFF && p.a_scale != nullptr;
#define NSID_GEMM_GO(HH, SS)                                                                              \
  do {                                                                                                    \
    if (aff) NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, HH, SS, CAN_AFF>), grid, dim3(256), 0, s, p);        \
    else NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, HH, SS, false>), grid, dim3(256), 0, s, p);              \
  } while (0)
  // FULL: whole tiles and an even number of whole stages per split -> the predication-free instantiation
  const bool full = st16 && p.I % BM == 0 && p.J % BN == 0 && p.rchunk % (2 * bk) == 0 && p.R % p.rchunk == 0;
  // Deep pipeline (KS = 2: 64-deep stages; PD = 4: four stages in flight) for launches that put at most ~2 workgroups on a
  // CU: there the round-1 loop kept 16-32 KB per workgroup in flight and fetched at ~16 GB/s per CU (shape table of round 2:
  // the 256-tile GEMMs of the C = 256 stage ran 32 us against a 7 us HBM/MFMA bound). Tuning keys gemm_deep_{ks,pd,max_wg,ec,kinds}.
  // Measured on MI355X with cold operands (tools/gemm_bench.py --cold, round 2), 256-512 workgroups per launch:
  //   forward        : KS = 2, two register sets, early commit with the interleaved schedule: 26.4 -> 18.2 us (16384x256x1024),
  //                    44.5 -> 27.0 us (8192x512x2048), 15.7 -> 11.9 us (16384x256x512);
  //   weight gradient: KS = 2 without early commit: 16.4 -> 12.9 us (16384x256x256), 20.8 -> 18.7 us; early commit loses 30 %;
  //   backward-data  : no deep form wins (the epilogue with addend / BatchNorm sums dominates): round-1 loop kept.
  // In the TWO-STREAM training step the isolated gains mostly vanish: a workgroup that owns a CU's LDS (120 KB with three
  // stage buffers) keeps the other view's kernels off that CU, and what the step rewards is little resource-time per
  // tile, not latency. One-box A/B of the whole step (two repetitions each): round-1 loops 8.52 / 8.52 ms, KS = 2 for
  // forward + weight gradient 8.41 / 8.43 ms, the same with early commit 8.55 / 8.57 ms -> KS = 2 without early commit is the
  // default; early commit (tuning key gemm_deep_ec = 1) remains for single-stream use (inference, microbenchmarks).
  const int deep_ks = (int)nsid_tune(NSID_T_gemm_deep_ks), deep_pd = (int)nsid_tune(NSID_T_gemm_deep_pd);
  const long deep_maxwg = nsid_tune(NSID_T_gemm_deep_max_wg);
  const int deep_ec = (int)nsid_tune(NSID_T_gemm_deep_ec);
  // which GEMM kinds take the deep form: bit 0 forward, bit 1 backward-data, bit 2 weight gradient
  const int deep_kinds = (int)nsid_tune(NSID_T_gemm_deep_kinds);
  constexpr int kind_bit = AR ? (BR ? 1 : 2) : 4;
  const long wgs = (long)tiles * p.rsplit * groups;
  int ks = 1, pd = 0;
  if (NW == 4 && full && wgs <= deep_maxwg && (deep_ks > 1 || deep_pd > 2) && (deep_kinds & kind_bit)) {
    ks = deep_ks == 2 ? 2 : 1;
    pd = deep_pd == 4 ? 4 : (ks == 2 ? 2 : 0);
    const int need = 32 * ks * (pd ? pd : 2);                     // whole register-set rounds of whole stages
    if (p.rchunk % need != 0) {                                   // fall back one notch at a time
      if (pd == 4 && p.rchunk % (32 * ks * 2) == 0) pd = ks == 2 ? 2 : 0;
      else if (ks == 2 && p.rchunk % (32 * (pd ? pd : 2)) == 0) ks = 1;
      else { ks = 1; pd = 0; }
      if (ks == 1 && pd == 2) pd = 0;
    }
  }
  const bool ec = deep_ec != 0 && BM * BN <= 128 * 128 && (AR && BR);   // forward only; three buffers of a 256x128 tile do not fit LDS
  if (p.abn_r != nullptr) {     // outside the fused BatchNorm-backward form: 1 = nothing launched, the caller runs the unfused pair
    if (!(AR && !BR && NW == 4 && BM == 128)) return NSID_EINVAL;
    if (!(st16 && w_bf16 && full)) return 1;
  }
  nsid_count(AR ? (BR ? NSID_C_gemm_fwd : NSID_C_gemm_bwd_data) : NSID_C_gemm_bwd_weight);
  if (p.split_major) nsid_count(NSID_C_gemm_split_major);
  if (aff) nsid_count(NSID_C_gemm_affine_load);
  if (AR && !BR && p.bn_r != nullptr) nsid_count(NSID_C_gemm_bn_sums);
  if (!AR) nsid_count(BM == BN ? NSID_C_wgrad_square : NSID_C_wgrad_rect);
  {
    // the deep / full-tile forms exist for bf16 storage only (and bf16 weights in the forward / backward-data kinds)
    const bool deep_path = st16 && full && (AR ? w_bf16 : true);
    if (deep_path) {
      nsid_count(NSID_C_gemm_full);
      if (NW == 4 && ks == 2) nsid_count(NSID_C_gemm_ks2);
      if (NW == 4 && pd == 4) nsid_count(NSID_C_gemm_pd4);
      if (NW == 4 && ec && (ks == 2 || pd == 4)) nsid_count(NSID_C_gemm_ec);
    }
  }
#define NSID_GEMM_DEEP_GO(AFF_, WB_, RELU_)                                                                       \
  do {                                                                                                            \
    if constexpr (NW == 8) {                                                                                      \
      NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, AFF_, WB_, true, RELU_, 1, 0, false, 0, 8>), grid, dim3(512), 0, s, p); \
      break;                                                                                                      \
    }                                                                                                             \
    if constexpr (BM * BN <= 128 * 128) {                                                                         \
      if (ec && ks == 2 && pd == 4) { NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, AFF_, WB_, true, RELU_, 2, 4, true>), grid, dim3(256), 0, s, p); break; } \
      if (ec && ks == 2) { NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, AFF_, WB_, true, RELU_, 2, 2, true>), grid, dim3(256), 0, s, p); break; }            \
      if (ec && pd == 4) { NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, AFF_, WB_, true, RELU_, 1, 4, true>), grid, dim3(256), 0, s, p); break; }            \
    }                                                                                                             \
    if (ks == 2 && pd == 4) NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, AFF_, WB_, true, RELU_, 2, 4>), grid, dim3(256), 0, s, p); \
    else if (ks == 2) NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, AFF_, WB_, true, RELU_, 2, 2>), grid, dim3(256), 0, s, p);       \
    else if (pd == 4) NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, AFF_, WB_, true, RELU_, 1, 4>), grid, dim3(256), 0, s, p);       \
    else NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, AFF_, WB_, true, RELU_>), grid, dim3(256), 0, s, p);                           \
  } while (0)
  if (p.abn_r != nullptr) {     // backward-data with the BatchNorm backward on the operand load: one full-tile bf16 form
    if constexpr (AR && !BR && NW == 4 && BM == 128) {
      nsid_count(NSID_C_gemm_bn_apply_load);
      NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, false, true, true, false, 1, 0, false, 0, 4, true>), grid, dim3(256), 0, s, p);
      return nsid_launch_status();
    } else {
      return NSID_EINVAL;
    }
  }
  if constexpr (NW == 8) {
    if (!(st16 && full && (w_bf16 || !AR))) return NSID_EINVAL;      // the 8-wave form exists for full bf16 tiles only
  }
  if constexpr (AR) {
    if (st16 && w_bf16) {
      const bool act_only = CAN_AFF && p.a_scale == nullptr && p.a_slope != 1.f;     // activation on load, no affine
      if (act_only) {
        if constexpr (CAN_AFF) {
          if (!full || p.a_slope != 0.f) return NSID_EINVAL;          // only ReLU on full tiles (the caller checks: ops.py)
          NSID_GEMM_DEEP_GO(false, true, true);
        }
      } else if (full) {
        if (aff) NSID_GEMM_DEEP_GO(CAN_AFF, true, false);
        else NSID_GEMM_DEEP_GO(false, true, false);
      } else {
        if (aff) NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, CAN_AFF, true>), grid, dim3(256), 0, s, p);
        else NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, false, true>), grid, dim3(256), 0, s, p);
      }
      return nsid_launch_status();
    }
  } else {
    if (st16 && full) {            // weight gradient, bf16 operands
      NSID_GEMM_DEEP_GO(false, false, false);
      return nsid_launch_status();
    }
  }
#undef NSID_GEMM_DEEP_GO
  if (w_bf16) return NSID_EINVAL;
  if (st16) NSID_GEMM_GO(true, true);
  else if (half) NSID_GEMM_GO(true, false);
  else NSID_GEMM_GO(false, false);
#undef NSID_GEMM_GO
  return nsid_launch_status();
}

// padded strided views (PADX = 1: left operand, 2: right operand): predicated instantiations only, every arithmetic mode
template <int BM, int BN, bool AR, bool BR, int PADX>
int launch_pad(GemmArgs p, int groups, hipStream_t s, int act_dtype, bool w_bf16) {
  const int tiles = ((p.I + BM - 1) / BM) * ((p.J + BN - 1) / BN);
  const bool st16 = act_dtype == NSID_BF16;
  const bool half = g_gemm_precision == NSID_GEMM_BF16 || st16;
  const int bk = half ? 32 : 16;
  p.rchunk = (p.rchunk + bk - 1) / bk * bk;
  p.rsplit = (p.R + p.rchunk - 1) / p.rchunk;
  p.split_major = p.rsplit > 1 && (p.rsplit % 8) == 0;
  dim3 grid(p.split_major ? p.rsplit : tiles, p.split_major ? tiles : p.rsplit, groups);
  if (w_bf16 && !(st16 && AR)) return NSID_EINVAL;
  if (st16 && w_bf16) {
    if constexpr (AR)
      NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, false, true, false, false, 1, 0, false, PADX>), grid, dim3(256), 0, s, p);
  } else if (st16) {
    NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, true, false, false, false, false, 1, 0, false, PADX>), grid, dim3(256), 0, s, p);
  } else if (half) {
    NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, true, false, false, false, false, false, 1, 0, false, PADX>), grid, dim3(256), 0, s, p);
  } else {
    NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR, false, false, false, false, false, false, 1, 0, false, PADX>), grid, dim3(256), 0, s, p);
  }
  return nsid_launch_status();
}

}  // namespace

extern "C" int nsid_version(void) { return 5; }

// ---- Downsample (Conv2d 3x3 stride 2 pad 1 on a width-1 map, encoder/graph_encoder.py:44) WITHOUT im2col -------------------
// Only kernel column 1 meets data: out[b*No + n'] = sum_t x[b*N + 2n'-1+t] . W_t (t = 0,1,2; row -1 of a clip is padding).
// With N even, x row-major and wp[o][t*C + c] = w[o][c][t][1], the im2col matrix is a VIEW of x:
//   col[m][kk] = xflat[(2m - 1)*C + kk],  kk in [0, 3C): row stride 2C, overlapping rows, base x - C,
// except that the first C columns of the rows m with m % No == 0 are the left padding (zero; they would alias the previous
// clip's last row). The three GEMMs below read that view through the padded-operand loads of gemm_kernel.
extern "C" int nsid_downsample3_fwd(const void* x, int B, int N, int C, const void* wp, int w_dtype, const float* bias,
                                    void* out, int Cout, float* stat, int act_dtype, void* stream) {
  NSID_REQUIRE(x && wp && out && B > 0 && N > 0 && N % 2 == 0 && C > 0 && Cout > 0 && NSID_DTYPE_OK(act_dtype));
  const int ch = act_dtype == NSID_BF16 ? 8 : 4;
  NSID_REQUIRE(C % ch == 0 && C % 4 == 0 && Cout % ch == 0 && Cout % 4 == 0 && nsid_aligned16(x) && nsid_aligned16(wp) &&
               nsid_aligned16(out) && NSID_DTYPE_OK(w_dtype) && (w_dtype == NSID_F32 || act_dtype == NSID_BF16));
  const int No = N / 2, M = B * No;
  const long esz = act_dtype == NSID_BF16 ? 2 : 4;
  GemmArgs p{};
  p.A = static_cast<const char*>(x) - (long)C * esz; p.lda = 2L * C; p.a_goff = 3L * C;
  p.B = wp; p.ldb = 3L * C; p.b_goff = (long)Cout * 3 * C;
  p.C = out; p.ldc = Cout; p.c_goff = Cout;
  p.I = M; p.J = Cout; p.R = 3 * C;
  p.a_slope = 1.f;
  p.bias = bias; p.bias_goff = Cout;
  p.stat = stat; p.stat_ld = Cout; p.stat_plane = (long)nsid_row_tiles(M) * Cout;
  p.rsplit = 1; p.rchunk = 3 * C;
  p.pad_period = No; p.pad_phase = 0; p.pad_c0 = 0; p.pad_c1 = C; p.pad_safe = C;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool wb = w_dtype == NSID_BF16;
  if (Cout <= 64) return launch_pad<128, 64, true, true, 1>(p, 1, s, act_dtype, wb);
  return launch_pad<128, 128, true, true, 1>(p, 1, s, act_dtype, wb);
}


// dwp[o][kk] += sum_m dout[m][o] * col[m][kk]   (fp32 atomics over row splits, like every weight gradient)
extern "C" int nsid_downsample3_bwd_weight(const void* dout, const void* x, float* dwp, int B, int N, int C, int Cout,
                                           int act_dtype, void* stream) {
  NSID_REQUIRE(dout && x && dwp && B > 0 && N > 0 && N % 2 == 0 && C > 0 && Cout > 0 && NSID_DTYPE_OK(act_dtype));
  const int ch = act_dtype == NSID_BF16 ? 8 : 4;
  NSID_REQUIRE(C % ch == 0 && Cout % ch == 0 && nsid_aligned16(dout) && nsid_aligned16(x));
  const int No = N / 2, M = B * No;
  const long esz = act_dtype == NSID_BF16 ? 2 : 4;
  GemmArgs p{};
  p.A = dout; p.lda = Cout; p.a_goff = Cout;                                   // A[R = m][i = o]
  p.B = static_cast<const char*>(x) - (long)C * esz; p.ldb = 2L * C; p.b_goff = 3L * C;   // B[R = m][j = kk]: the view
  p.C = dwp; p.ldc = 3L * C; p.c_goff = (long)Cout * 3 * C;
  p.I = Cout; p.J = 3 * C; p.R = M;
  p.b_slope = 1.f;
  p.atomic_out = 1;
  p.pad_period = No; p.pad_phase = 0; p.pad_c0 = 0; p.pad_c1 = C; p.pad_safe = C;
  const long tiles = (long)((Cout + 63) / 64) * ((3 * C + 63) / 64);
  long S = std::min<long>((M + 1023) / 1024, std::max<long>(1, 1024 / tiles));
  S = std::max<long>(S, (256 + tiles - 1) / tiles);
  S = std::min<long>(S, (M + 255) / 256);
  p.rsplit = (int)std::max<long>(S, 1);
  p.rchunk = (M + p.rsplit - 1) / p.rsplit;
  return launch_pad<64, 64, false, false, 2>(p, 1, static_cast<hipStream_t>(stream), act_dtype, false);
}

// dx[2n'] = dout[n'] . W_1 ; dx[2n'+1] = dout[n'] . W_2 + dout[n'+1] . W_0 (n'+1 inside the clip).
// w_even = wp + C (row stride 3C: the tap-1 columns), w_odd = [W_2 ; W_0] stacked as (2*Cout, C) (nsid_pack_ds_weight_bwd).
// The odd rows are ONE GEMM over overlapping rows of dout: A[m][0:2Cout] = doutflat[m*Cout : m*Cout + 2Cout], with the
// second half masked on the last node of every clip.
extern "C" int nsid_downsample3_bwd_data(const void* dout, const void* wp, const void* w_odd, int w_dtype, void* dx, int B,
                                         int N, int C, int Cout, int act_dtype, void* stream) {
  NSID_REQUIRE(dout && wp && w_odd && dx && B > 0 && N > 0 && N % 2 == 0 && C > 0 && Cout > 0 && NSID_DTYPE_OK(act_dtype));
  const int ch = act_dtype == NSID_BF16 ? 8 : 4;
  NSID_REQUIRE(C % ch == 0 && Cout % ch == 0 && nsid_aligned16(dout) && nsid_aligned16(wp) && nsid_aligned16(w_odd) &&
               nsid_aligned16(dx) && NSID_DTYPE_OK(w_dtype) && (w_dtype == NSID_F32 || act_dtype == NSID_BF16));
  const int No = N / 2, M = B * No;
  const long esz = act_dtype == NSID_BF16 ? 2 : 4, wsz = w_dtype == NSID_BF16 ? 2 : 4;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool wb = w_dtype == NSID_BF16;
  GemmArgs p{};
  p.A = dout; p.lda = Cout; p.a_goff = Cout;
  p.B = static_cast<const char*>(wp) + (long)C * wsz; p.ldb = 3L * C; p.b_goff = 0;      // B[R = o][j = c] = wp[o][C + c]
  p.C = dx; p.ldc = 2L * C; p.c_goff = C;
  p.I = M; p.J = C; p.R = Cout;
  p.rsplit = 1; p.rchunk = Cout;
  p.bn_slope = 1.f;
  int rc = C <= 64 ? launch<128, 64, true, false>(p, 1, s, act_dtype, wb) : launch<128, 128, true, false>(p, 1, s, act_dtype, wb);
  if (rc != NSID_OK) return rc;
  GemmArgs q{};
  q.A = dout; q.lda = Cout; q.a_goff = 2L * Cout;
  q.B = w_odd; q.ldb = C; q.b_goff = 0;
  q.C = static_cast<char*>(dx) + (long)C * esz; q.ldc = 2L * C; q.c_goff = C;
  q.I = M; q.J = C; q.R = 2 * Cout;
  q.rsplit = 1; q.rchunk = 2 * Cout;
  q.bn_slope = 1.f;
  q.pad_period = No; q.pad_phase = No - 1; q.pad_c0 = Cout; q.pad_c1 = 2 * Cout; q.pad_safe = 0;
  return C <= 64 ? launch_pad<128, 64, true, false, 1>(q, 1, s, act_dtype, wb)
                 : launch_pad<128, 128, true, false, 1>(q, 1, s, act_dtype, wb);
}

void* g_gemm_trace_host = nullptr;     // the same buffer for kernels of other translation units (gemm256.hip takes it as an argument)
extern "C" int nsid_debug_gemm_trace(void* buf) {
  g_gemm_trace_host = buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &buf, sizeof(buf)) == hipSuccess ? NSID_OK : NSID_EINVAL;
}
extern "C" int nsid_set_gemm_precision(int mode) {
  if (mode != NSID_GEMM_FP32 && mode != NSID_GEMM_BF16) return NSID_EINVAL;
  g_gemm_precision = mode;
  return NSID_OK;
}
extern "C" int nsid_get_gemm_precision(void) { return g_gemm_precision; }
extern "C" long nsid_gemm_g256_launches(void) { return g_g256_launches; }
extern "C" int nsid_row_tiles(int M) { return (M + NSID_ROW_TILE - 1) / NSID_ROW_TILE; }

static int linear_fwd_impl(const void* x, int ldx, const void* w, int w_dtype, const float* bias, void* out, int ldo,
                           int M, int Nout, int K, int groups, const float* in_scale, const float* in_shift, int act_in,
                           int act_out, float* stat, int ksplit, int act_dtype, const void* addend, int ldadd,
                           void* stream);

extern "C" int nsid_linear_fwd(const void* x, int ldx, const void* w, int w_dtype, const float* bias, void* out, int ldo,
                               int M, int Nout, int K, int groups, const float* in_scale, const float* in_shift,
                               int act_in, int act_out, float* stat, int ksplit, int act_dtype, void* stream) {
  return linear_fwd_impl(x, ldx, w, w_dtype, bias, out, ldo, M, Nout, K, groups, in_scale, in_shift, act_in, act_out, stat,
                         ksplit, act_dtype, nullptr, 0, stream);
}

// out = f(x) W^T + bias + addend: the residual stream added in the GEMM epilogue. With an eval-mode BatchNorm folded into
// (W, bias) this is a whole "conv + BN + shortcut" in one launch (functional.py, eval path).
extern "C" int nsid_linear_fwd_res(const void* x, int ldx, const void* w, int w_dtype, const float* bias,
                                   const void* addend, int ldadd, void* out, int ldo, int M, int Nout, int K, int groups,
                                   const float* in_scale, const float* in_shift, int act_in, int act_dtype,
                                   void* stream) {
  const int ch = act_dtype == NSID_BF16 ? 8 : 4;
  NSID_REQUIRE(addend && ldadd % ch == 0 && ldadd >= groups * Nout && nsid_aligned16(addend));
  return linear_fwd_impl(x, ldx, w, w_dtype, bias, out, ldo, M, Nout, K, groups, in_scale, in_shift, act_in,
                         NSID_ACT_NONE, nullptr, 1, act_dtype, addend, ldadd, stream);
}

static int linear_fwd_impl(const void* x, int ldx, const void* w, int w_dtype, const float* bias, void* out, int ldo,
                           int M, int Nout, int K, int groups, const float* in_scale, const float* in_shift, int act_in,
                           int act_out, float* stat, int ksplit, int act_dtype, const void* addend, int ldadd,
                           void* stream) {
  NSID_REQUIRE(x && w && out && M > 0 && Nout > 0 && K > 0 && groups > 0 && ksplit >= 1 && NSID_DTYPE_OK(act_dtype));
  NSID_REQUIRE(NSID_DTYPE_OK(w_dtype) && (w_dtype == NSID_F32 || (act_dtype == NSID_BF16 && K % 8 == 0)));
  const int ch = act_dtype == NSID_BF16 ? 8 : 4;     // elements per 16-byte chunk of the activation tensors
  NSID_REQUIRE(K % ch == 0 && ldx % ch == 0 && nsid_aligned16(x) && nsid_aligned16(w) && K % 4 == 0);
  NSID_REQUIRE(act_dtype == NSID_F32 || (ksplit == 1 && (act_out == NSID_ACT_NONE || act_out == NSID_ACT_RELU) && Nout % 8 == 0 && ldo % 8 == 0));
  NSID_REQUIRE(act_out != NSID_ACT_RELU || (ksplit == 1 && stat == nullptr && addend == nullptr));
  // ldx < K (overlapping rows of x, read-only) is allowed for one group: the STFT front end frames a waveform that way
  NSID_REQUIRE((ldx >= groups * K || (groups == 1 && ldx > 0)) && ldo >= groups * Nout);
  NSID_REQUIRE(Nout % 4 == 0 && ldo % 4 == 0 && nsid_aligned16(out) && (bias == nullptr || nsid_aligned16(bias)));
  NSID_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  NSID_REQUIRE(ksplit == 1 || (stat == nullptr && (act_out == NSID_ACT_NONE || (act_out == NSID_ACT_ELU && act_dtype == NSID_F32))));   // ELU is a separate pass over the finished sums (below)
  NSID_REQUIRE(act_in != NSID_ACT_ELU && (act_out == NSID_ACT_NONE || act_out == NSID_ACT_ELU || act_out == NSID_ACT_RELU));
  // an activation on load WITHOUT an affine exists as ReLU on bf16 operands with bf16 weights (full tiles; eval path)
  NSID_REQUIRE(in_scale != nullptr || act_in == NSID_ACT_NONE ||
               (act_in == NSID_ACT_RELU && act_dtype == NSID_BF16 && w_dtype == NSID_BF16));
  GemmArgs p{};
  p.A = x; p.lda = ldx; p.a_goff = K;
  p.B = w; p.ldb = K; p.b_goff = (long)Nout * K;
  p.C = out; p.ldc = ldo; p.c_goff = Nout;
  p.I = M; p.J = Nout; p.R = K;
  p.a_scale = in_scale; p.a_shift = in_shift; p.a_slope = act_slope(act_in); p.a_aff_goff = K;
  p.bias = bias; p.bias_goff = Nout;
  p.addend = addend; p.ldadd = ldadd;
  p.stat = stat; p.stat_ld = (long)groups * Nout; p.stat_plane = (long)nsid_row_tiles(M) * groups * Nout;
  p.rsplit = ksplit;
  p.rchunk = (K + ksplit - 1) / ksplit;
  p.atomic_out = ksplit > 1;
  p.relu_out = act_out == NSID_ACT_RELU;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the statistics tile must be NSID_ROW_TILE rows, so BM = 128 always; narrow outputs take the 64-column tile
  // bf16 operands make the kernel latency/HBM-bound: when 128-wide tiles would give fewer than two workgroups per CU,
  // halve the tile width to double the loads in flight (the statistics tile stays 128 rows)
  const long t128 = (long)nsid_row_tiles(M) * ((Nout + 127) / 128) * groups;
  const bool half = g_gemm_precision == NSID_GEMM_BF16 || act_dtype == NSID_BF16;
  const int force_narrow = (int)nsid_tune(NSID_T_fwd_narrow);
  // measured (tools/gemm_bench.py, bf16 path): the 128-wide tile wins from 256 output columns on, and always when the
  // left operand carries the producer's BatchNorm (every column tile re-applies it: fewer, wider tiles = less VALU)
  bool narrow = half ? (Nout <= 64 || (Nout <= 128 && K <= 256 && in_scale == nullptr)) : Nout <= 64;
  if (half && t128 < 128) narrow = true;        // few row tiles (the projector head, M = batch): more, narrower workgroups
  if (force_narrow >= 0 && Nout > 64) narrow = force_narrow != 0;
  const bool wb = w_dtype == NSID_BF16;
  // (256x128-tile forms of this kernel -- 4 waves "tall", 8 waves "w8" -- existed through round 3: every instantiation spilled 13-81
  // registers to scratch, and the extraction bench measured them within 0.3 % of the 128x128 tiles (4.540 against 4.555 ms per
  // micro-batch, docs/experiments.md round 4): removed. Large forward-only shapes go to gemm256.hip below.)
  // 256x256 tiles with LDS-DMA staging (gemm256.hip): tuning key g256_min = smallest number of 256x256 tiles that takes it (0 = never)
  // default 512 (two tiles per CU and more: fingerprinting at micro-batch 2 048): 7.09 -> 6.56 ms per micro-batch. The training step
  // (<= 256 such tiles per launch) is neutral to slightly worse with it (8.30 vs 8.34 ms: a workgroup that owns 150 KB of a CU's LDS
  // keeps the other view's kernels off that CU), so it stays on gemm.hip.
  // weight-stationary streaming form (wsgemm.hip) for the layers whose whole weight matrix fits LDS: tuning key ws_gemm bit 0
  if ((nsid_tune(NSID_T_ws_gemm) & 1) && act_dtype == NSID_BF16 && wb && ksplit == 1 && act_out == NSID_ACT_NONE && addend == nullptr &&
      act_in != NSID_ACT_ELU) {
    const int rcw = nsid_ws_fwd_launch(x, ldx, w, bias, out, ldo, M, Nout, K, groups, in_scale, in_shift, act_slope(act_in), stat,
                                       p.stat_plane, p.stat_ld, s);
    if (rcw != 1) { nsid_count(NSID_C_ws_fwd); return rcw; }
  }
  const long g256_min = nsid_tune(NSID_T_g256_min);
  if (g256_min > 0 && (stat == nullptr || nsid_tune(NSID_T_g256_train) != 0) && act_dtype == NSID_BF16 && wb && groups == 1 && in_scale == nullptr && ksplit == 1 &&
      (act_out == NSID_ACT_NONE || act_out == NSID_ACT_RELU) && act_in == NSID_ACT_NONE && ldx >= K &&
      (long)(M / 256) * (Nout / 256) >= g256_min) {
    const int rc256 = nsid_gemm256_fwd_launch(x, ldx, w, bias, addend, ldadd, out, ldo, M, Nout, K, act_out == NSID_ACT_RELU, stat,
                                              p.stat_plane, p.stat_ld, s);
    if (rc256 != 1) { ++g_g256_launches; nsid_count(NSID_C_gemm256); return rc256; }
  }
  const int rc = narrow ? launch<128, 64, true, true>(p, groups, s, act_dtype, wb) : launch<128, 128, true, true>(p, groups, s, act_dtype, wb);
  if (rc != NSID_OK || act_out != NSID_ACT_ELU) return rc;
  const long n = (long)M * groups * Nout;
  NSID_LAUNCH(elu_inplace_kernel, dim3((int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, s,
              static_cast<float*>(out), (long)M, groups * Nout, (long)ldo);
  return nsid_launch_status();
}

struct AbnArgs {          // BatchNorm backward of the layer in front, applied on the operand load (GemmArgs::abn_*)
  const void* r; const float* coef4; int act; void* dr;
};
static int linear_bwd_data_impl(const void* dout, int ldd, const void* w, int w_dtype, const void* addend, int ldadd,
                                void* din, int ldi, int M, int Nout, int K, int groups, int act_dtype, void* stream,
                                const void* bn_r, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                const float* bn_invstd, int bn_act, float* bn_partial, const AbnArgs* abn = nullptr);

extern "C" int nsid_linear_bwd_data(const void* dout, int ldd, const void* w, int w_dtype, const void* addend, int ldadd,
                                    void* din, int ldi, int M, int Nout, int K, int groups, int act_dtype,
                                    void* stream) {
  return linear_bwd_data_impl(dout, ldd, w, w_dtype, addend, ldadd, din, ldi, M, Nout, K, groups, act_dtype, stream,
                              nullptr, nullptr, nullptr, nullptr, nullptr, NSID_ACT_NONE, nullptr);
}

extern "C" int nsid_linear_bwd_data_bn(const void* dout, int ldd, const void* w, int w_dtype, const void* addend,
                                       int ldadd, void* din, int ldi, int M, int Nout, int K, int groups, int act_dtype,
                                       const void* bn_r, const float* bn_scale, const float* bn_shift,
                                       const float* bn_mean, const float* bn_invstd, int bn_act, float* bn_partial,
                                       void* stream) {
  NSID_REQUIRE(bn_r && bn_scale && bn_shift && bn_mean && bn_invstd && bn_partial && act_dtype == NSID_BF16);
  NSID_REQUIRE(ldi == groups * K && (groups * K) % 8 == 0 && nsid_aligned16(bn_r));
  NSID_REQUIRE(bn_act == NSID_ACT_NONE || bn_act == NSID_ACT_RELU || bn_act == NSID_ACT_LEAKY);
  return linear_bwd_data_impl(dout, ldd, w, w_dtype, addend, ldadd, din, ldi, M, Nout, K, groups, act_dtype, stream,
                              bn_r, bn_scale, bn_shift, bn_mean, bn_invstd, bn_act, bn_partial);
}

// Backward-data of a conv whose OUTPUT gradient still has to go through the BatchNorm(+activation) backward of that conv's own
// BatchNorm: din = addend + dr w with dr = BN-backward(dy, r) evaluated on the operand load (no bn_bwd_apply pass), dr written once as
// a side output for the weight gradient. coef4[4][groups*Nout] = {sc, sh, P, Q} from nsid_bn_bwd_finalize_fused. The optional bn_*
// arguments are nsid_linear_bwd_data_bn's (column sums for the NEXT BatchNorm backward). Returns 1 (nothing launched) when the shape is
// outside the fused form: bf16 storage and weights, M % 128 == 0, Nout % 64 == 0, K a multiple of the tile width, ldd == groups*Nout.
extern "C" int nsid_linear_bwd_data_bnapply(const void* dy, const void* r, const float* coef4, int act, void* dr, const void* w,
                                            int w_dtype, const void* addend, int ldadd, void* din, int ldi, int M, int Nout, int K,
                                            int groups, int act_dtype, const void* bn_r, const float* bn_scale,
                                            const float* bn_shift, const float* bn_mean, const float* bn_invstd, int bn_act,
                                            float* bn_partial, void* stream) {
  NSID_REQUIRE(dy && r && coef4 && dr && nsid_aligned16(r) && nsid_aligned16(dr) && nsid_aligned16(coef4));
  // only column tile 0 writes dr while every column tile of the row panel reads dy and r, unordered: an aliased call would race
  NSID_REQUIRE(dr != dy && dr != r);
  NSID_REQUIRE(act == NSID_ACT_NONE || act == NSID_ACT_RELU || act == NSID_ACT_LEAKY);
  if (act_dtype != NSID_BF16 || w_dtype != NSID_BF16 || (groups * Nout) % 8 != 0) return 1;
  if (bn_r != nullptr) {
    NSID_REQUIRE(bn_scale && bn_shift && bn_mean && bn_invstd && bn_partial && ldi == groups * K && (groups * K) % 8 == 0 &&
                 nsid_aligned16(bn_r));
    NSID_REQUIRE(bn_act == NSID_ACT_NONE || bn_act == NSID_ACT_RELU || bn_act == NSID_ACT_LEAKY);
  }
  const AbnArgs abn{r, coef4, act, dr};
  return linear_bwd_data_impl(dy, groups * Nout, w, w_dtype, addend, ldadd, din, ldi, M, Nout, K, groups, act_dtype, stream, bn_r,
                              bn_scale, bn_shift, bn_mean, bn_invstd, bn_act, bn_partial, &abn);
}

static int linear_bwd_data_impl(const void* dout, int ldd, const void* w, int w_dtype, const void* addend, int ldadd,
                                void* din, int ldi, int M, int Nout, int K, int groups, int act_dtype, void* stream,
                                const void* bn_r, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                const float* bn_invstd, int bn_act, float* bn_partial, const AbnArgs* abn) {
  NSID_REQUIRE(dout && w && din && M > 0 && Nout > 0 && K > 0 && groups > 0 && NSID_DTYPE_OK(act_dtype));
  NSID_REQUIRE(NSID_DTYPE_OK(w_dtype) && (w_dtype == NSID_F32 || (act_dtype == NSID_BF16 && K % 8 == 0)));
  const bool wb = w_dtype == NSID_BF16;
  const int ch = act_dtype == NSID_BF16 ? 8 : 4;
  NSID_REQUIRE(Nout % ch == 0 && K % ch == 0 && ldd % ch == 0 && ldi % ch == 0 && nsid_aligned16(dout) && nsid_aligned16(w));
  NSID_REQUIRE(addend == nullptr || ldadd % ch == 0);
  NSID_REQUIRE(ldd >= groups * Nout && ldi >= groups * K && ldi % 4 == 0 && nsid_aligned16(din));
  NSID_REQUIRE(addend == nullptr || (ldadd % 4 == 0 && ldadd >= groups * K && nsid_aligned16(addend)));
  GemmArgs p{};
  p.A = dout; p.lda = ldd; p.a_goff = Nout;
  p.B = w; p.ldb = K; p.b_goff = (long)Nout * K;     // B[R = n][j = k]
  p.C = din; p.ldc = ldi; p.c_goff = K;
  p.I = M; p.J = K; p.R = Nout;
  p.addend = addend; p.ldadd = ldadd;
  p.rsplit = 1; p.rchunk = Nout;
  p.bn_r = bn_r; p.bn_ldr = (long)groups * K;
  p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd;
  p.bn_slope = act_slope(bn_act);
  p.bn_partial = bn_partial; p.bn_ld = (long)groups * K; p.bn_plane = (long)nsid_row_tiles(M) * groups * K;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long t128 = (long)nsid_row_tiles(M) * ((K + 127) / 128) * groups;
  const bool half = g_gemm_precision == NSID_GEMM_BF16 || act_dtype == NSID_BF16;
  const int force_narrow = (int)nsid_tune(NSID_T_bwd_narrow);
  bool narrow = half ? (K <= 64 || (K <= 128 && Nout <= 256)) : K <= 64;
  if (half && t128 < 128) narrow = true;        // few row tiles (the projector head, M = batch)
  if (force_narrow >= 0 && K > 64) narrow = force_narrow != 0;
  // weight-stationary streaming form (wsgemm.hip): tuning key ws_gemm bit 1 (plain operand) / bit 2 (BatchNorm backward on the operand load)
  if ((nsid_tune(NSID_T_ws_gemm) & (abn ? 4 : 2)) && act_dtype == NSID_BF16 && wb &&
      (bn_r == nullptr || bn_partial != nullptr)) {
    const int rcw = nsid_ws_bwd_data_launch(dout, ldd, w, addend, ldadd, din, ldi, M, Nout, K, groups, bn_r, p.bn_ldr, bn_scale, bn_shift,
                                            bn_mean, bn_invstd, p.bn_slope, bn_partial, p.bn_plane, p.bn_ld, abn ? abn->r : nullptr,
                                            abn ? abn->coef4 : nullptr, (long)groups * Nout, abn ? act_slope(abn->act) : 1.f,
                                            abn ? abn->dr : nullptr, (long)groups * Nout, s);
    if (rcw != 1) { nsid_count(abn ? NSID_C_ws_bwd_bnapply : NSID_C_ws_bwd_data); return rcw; }
  }
  if (abn != nullptr) {
    // every column tile re-evaluates the BatchNorm backward of its row panel: as few column tiles as the tile family offers
    narrow = K <= 64;
    p.abn_r = abn->r; p.abn_coef = abn->coef4; p.abn_plane = (long)groups * Nout; p.abn_slope = act_slope(abn->act);
    p.abn_dr = abn->dr; p.abn_lddr = (long)groups * Nout;
    if (K % (narrow ? 64 : 128) != 0 || M % 128 != 0 || Nout % 64 != 0) return 1;
  }
  // A handful of tiles with a long reduction (the projector head: M = batch rows, Nout = 4096 -> K = 1024 was 32 workgroups of 64 stages
  // each, 61 us on the turnaround between forward and backward where nothing else runs): split the reduction over up to 256
  // workgroups, fp32 atomics into the zeroed output. fp32 storage, plain epilogue only (tuning key bwd_split_max_tiles, 0 = never).
  if (act_dtype == NSID_F32 && addend == nullptr && bn_r == nullptr && abn == nullptr && groups == 1 && Nout >= 1024 && ldi == K) {
    const long tiles = (long)nsid_row_tiles(M) * ((K + (narrow ? 63 : 127)) / (narrow ? 64 : 128));
    const long maxt = nsid_tune(NSID_T_bwd_split_max_tiles);
    const long S = maxt > 0 && tiles <= maxt ? std::min<long>(256 / tiles, Nout / 256) : 1;
    if (S > 1) {
      // (the library's own fill kernel: a memset node did not replay with the captured step)
      const int rcz = nsid_fill_zero(din, (size_t)M * K * 4, stream);
      if (rcz != NSID_OK) return rcz;
      p.rsplit = (int)S;
      p.rchunk = (int)((Nout + S - 1) / S);
      p.atomic_out = 1;
      nsid_count(NSID_C_gemm_bwd_split);
    }
  }
  if (narrow) return launch<128, 64, true, false>(p, groups, s, act_dtype, wb);
  return launch<128, 128, true, false>(p, groups, s, act_dtype, wb);
}

extern "C" int nsid_linear_bwd_weight(const void* dout, int ldd, const void* x, int ldx, float* dw, int M, int Nout,
                                      int K, int groups, const float* in_scale, const float* in_shift, int act_in,
                                      int act_dtype, void* stream) {
  NSID_REQUIRE(dout && x && dw && M > 0 && Nout > 0 && K > 0 && groups > 0 && NSID_DTYPE_OK(act_dtype));
  const int ch = act_dtype == NSID_BF16 ? 8 : 4;
  NSID_REQUIRE(Nout % ch == 0 && K % ch == 0 && ldd % ch == 0 && ldx % ch == 0 && nsid_aligned16(dout) && nsid_aligned16(x));
  NSID_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  GemmArgs p{};
  p.A = dout; p.lda = ldd; p.a_goff = Nout;          // A[R = m][i = n]
  p.B = x; p.ldb = ldx; p.b_goff = K;                // B[R = m][j = k]
  p.C = dw; p.ldc = K; p.c_goff = (long)Nout * K;
  p.I = Nout; p.J = K; p.R = M;
  p.b_scale = in_scale; p.b_shift = in_shift; p.b_slope = act_slope(act_in); p.b_aff_goff = K;
  p.atomic_out = 1;
  // Tile and split choice (measured on MI355X, tools/gemm_bench.py): every workgroup ends in a tile-sized burst of fp32
  // atomics that all XCDs resolve memory-side (~1.3 TB/s chip-wide), so the bytes of atomics = workgroups x tile bytes
  // decide the kernel: 64x64 tiles (16 KB) beat 128x128 (64 KB) at every shape of the encoder once >= 256 workgroups
  // are in flight. Splits: >= 1024 rows each, at most ~1024 workgroups, at least ~256.
  const bool use_wide = nsid_tune(NSID_T_wgrad_wide) != 0;
  if (use_wide && act_dtype == NSID_BF16 && nsid_aligned16(dout) && nsid_aligned16(x) &&
      (in_scale == nullptr || (nsid_aligned16(in_scale) && nsid_aligned16(in_shift)))) {
    const int rc = nsid_wgrad2_launch(dout, ldd, x, ldx, dw, M, Nout, K, groups, in_scale, in_shift, act_slope(act_in),
                                      static_cast<hipStream_t>(stream));
    if (rc != 1) return rc;                  // 1 = shape outside the fast form's preconditions
  }
  // 128x64 tiles (a quarter fewer operand bytes per flop than 64x64) where they still give >= 256 workgroups of >= 1024
  // rows, i.e. without more splits = atomic bytes than the square tiles need
  // (a launch alone is ~8 % slower with them, the two-branch step 1 % faster: half the workgroups and LDS per CU leave the
  // other view's kernels more room — judged by one-box A/B of the whole step)
  const long rect = nsid_tune(NSID_T_wgrad_rect);
  const long wg_rect = nsid_tune(NSID_T_wgrad_wgs_rect);    // workgroup targets: rect 512 -> 256 is worth
  // 0.12 ms of the two-stream step (8.23 -> 8.11 ms, one-box A/B x2; 192 / 128 fall back to square tiles and lose 0.2 ms): half the splits =
  // half the atomic bytes, and a weight gradient is off the critical chain, so its own latency does not matter
  const long wg_sq = nsid_tune(NSID_T_wgrad_wgs_sq);
  if (rect && act_dtype == NSID_BF16 && Nout % 128 == 0 && K % 64 == 0 && M % 1024 == 0) {
    const long tiles_r = (long)(Nout / 128) * (K / 64) * groups;
    const long S = std::min<long>(M / 1024, std::max<long>(1, wg_rect / tiles_r));
    const long rect_min = nsid_tune(NSID_T_wgrad_rect_min);
    if (tiles_r * S >= rect_min) {
      p.rsplit = (int)S;
      p.rchunk = (int)(M / S);
      return launch<128, 64, false, false>(p, groups, static_cast<hipStream_t>(stream), act_dtype);
    }
  }
  // fp32 arithmetic (16x lower matrix rate) stays MFMA-bound: there the larger tile wins whenever it fills the chip.
  const bool half = g_gemm_precision == NSID_GEMM_BF16 || act_dtype == NSID_BF16;
  const long t128 = (long)((Nout + 127) / 128) * ((K + 127) / 128) * groups;
  const bool small = half || (Nout <= 64 || K <= 64) || t128 * ((M + 511) / 512) < 256;
  const int bm = small ? 64 : 128;
  const long tiles = (long)((Nout + bm - 1) / bm) * ((K + bm - 1) / bm) * groups;
  long S;
  if (half) {
    S = std::min<long>((M + 1023) / 1024, std::max<long>(1, wg_sq / tiles));
    const long sq_min = nsid_tune(NSID_T_wgrad_sq_min);
    S = std::max<long>(S, (sq_min + tiles - 1) / tiles);
    S = std::min<long>(S, (M + 255) / 256);
  } else {
    S = std::min<long>((512 + tiles - 1) / tiles, (M + 511) / 512);
  }
  int rsplit = (int)std::max<long>(S, 1);
  p.rsplit = rsplit;
  p.rchunk = (M + rsplit - 1) / rsplit;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (small) return launch<64, 64, false, false>(p, groups, s, act_dtype);
  return launch<128, 128, false, false>(p, groups, s, act_dtype);
}

// Many weight gradients in one launch per tile class (see wgrad_grouped_kernel). problems[i].dout[1] / x[1] NULL: one row segment.
extern "C" int nsid_linear_bwd_weight_grouped(const nsid_wgrad_problem* problems, int n, int act_dtype, int max_workgroups,
                                              void* stream) {
  NSID_REQUIRE(problems && n > 0 && NSID_DTYPE_OK(act_dtype) && max_workgroups >= 0);
  const bool st16 = act_dtype == NSID_BF16;
  const int ch = st16 ? 8 : 4;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long rows_target = std::max<long>(128, nsid_tune(NSID_T_wgg_rows));
  const long rows_sq = std::max<long>(128, nsid_tune(NSID_T_wgg_rows_sq)), rows_gen = std::max<long>(32, nsid_tune(NSID_T_wgg_rows_gen));
  const bool use_w3 = nsid_tune(NSID_T_wgg_w3) != 0;
  // classes: 0 = 128x64 full tiles (64-deep stages), 1 = 64x64 full tiles, 2 = 64x64 predicated (any shape),
  //          3 / 4 = 128x128 tiles, 8 waves (wgrad.hip), without / with the producer affine on x,
  //          5 = Downsample (64x64 predicated over the padded 3-tap view of x), 6 = fp32 storage (64x64 predicated; the projector head)
  constexpr int NCLS = 7;
  WgGroupArgs ga[NCLS];
  long wgs[NCLS] = {0, 0, 0, 0, 0, 0, 0};
  const bool half = g_gemm_precision == NSID_GEMM_BF16 || st16;
  for (int c = 0; c < NCLS; ++c) ga[c].n = 0;
  auto flush = [&](int c) -> int {
    if (ga[c].n == 0) return NSID_OK;
    ga[c].wg0[ga[c].n] = (int)wgs[c];
    long gsz = wgs[c];
    if (max_workgroups > 0 && gsz > max_workgroups) gsz = std::max(8, max_workgroups / 8 * 8);     // (a multiple of 8: item w stays on XCD w % 8)
    const dim3 grid((unsigned)gsz), block(256);
    if (c == 3 || c == 4) {
      nsid_count(NSID_C_wgrad_grouped);
      nsid_count(NSID_C_wgrad_grouped_w3);
      const int rc3 = nsid_wgrad3_grouped_launch(ga[c], (int)gsz, c == 4, s);
      ga[c].n = 0;
      wgs[c] = 0;
      return rc3;
    }
    if (c == 0) NSID_LAUNCH((wgrad_grouped_kernel<128, 64, true>), grid, block, 0, s, ga[c]);
    else if (c == 1) NSID_LAUNCH((wgrad_grouped_kernel<64, 64, true>), grid, block, 0, s, ga[c]);
    else if (c == 2) NSID_LAUNCH((wgrad_grouped_kernel<64, 64, false>), grid, block, 0, s, ga[c]);
    else if (c == 5) NSID_LAUNCH((wgrad_grouped_kernel<64, 64, false, true, true, 2>), grid, block, 0, s, ga[c]);
    else if (half) NSID_LAUNCH((wgrad_grouped_kernel<64, 64, false, true, false>), grid, block, 0, s, ga[c]);
    else NSID_LAUNCH((wgrad_grouped_kernel<64, 64, false, false, false>), grid, block, 0, s, ga[c]);
    nsid_count(NSID_C_wgrad_grouped);
    ga[c].n = 0;
    wgs[c] = 0;
    return nsid_launch_status();
  };
  for (int i = 0; i < n; ++i) {
    const nsid_wgrad_problem& q = problems[i];
    const int nseg = q.dout[1] != nullptr ? 2 : 1;
    NSID_REQUIRE(q.dout[0] && q.x[0] && q.dw && q.M > 0 && q.Nout > 0 && q.K > 0 && q.groups > 0);
    NSID_REQUIRE((q.dout[1] == nullptr) == (q.x[1] == nullptr));
    NSID_REQUIRE(q.Nout % ch == 0 && q.K % ch == 0 && q.ldd % ch == 0 && q.ldx % ch == 0 && q.ds_out_nodes >= 0);
    const bool ds = q.ds_out_nodes > 0;       // x[v]: the (B * 2 * ds_out_nodes, K / 3) input of a Downsample; dw: its packed (Nout, K) gradient
    NSID_REQUIRE(!ds || (st16 && q.groups == 1 && q.K % 3 == 0 && (q.K / 3) % 8 == 0 && q.M % q.ds_out_nodes == 0 && q.in_scale[0] == nullptr &&
                         q.ldx == q.K / 3));
    for (int v = 0; v < nseg; ++v) {
      NSID_REQUIRE(nsid_aligned16(q.dout[v]) && nsid_aligned16(q.x[v]));
      NSID_REQUIRE((q.in_scale[v] == nullptr) == (q.in_shift[v] == nullptr));
      NSID_REQUIRE((q.in_scale[v] == nullptr) == (q.in_scale[0] == nullptr));       // both segments with or without the affine
      NSID_REQUIRE(q.in_scale[v] == nullptr || (nsid_aligned16(q.in_scale[v]) && nsid_aligned16(q.in_shift[v])));
    }
    const bool rows_ok = q.M % 128 == 0;
    int cls = (rows_ok && q.Nout % 128 == 0 && q.K % 64 == 0) ? 0 : ((rows_ok && q.Nout % 64 == 0 && q.K % 64 == 0) ? 1 : 2);
    if (cls == 0 && use_w3 && q.K % 128 == 0) cls = q.in_scale[0] != nullptr ? 4 : 3;
    if (ds) cls = 5;
    if (!st16) cls = 6;
    const int bm = (cls == 0 || cls == 3 || cls == 4) ? 128 : 64, bn = (cls == 3 || cls == 4) ? 128 : 64;
    const long rows_cls = (cls == 1 || cls == 5) ? rows_sq : ((cls == 2 || cls == 6) ? rows_gen : rows_target);
    WgProb w{};
    for (int v = 0; v < 2; ++v) {
      w.A[v] = q.dout[v]; w.B[v] = q.x[v]; w.bsc[v] = q.in_scale[v]; w.bsh[v] = q.in_shift[v];
    }
    w.C = q.dw; w.lda = q.ldd; w.ldb = q.ldx; w.I = q.Nout; w.J = q.K; w.R = q.M; w.groups = q.groups;
    if (ds) {       // the im2col matrix as a view of x: col[m][kk] = xflat[(2m - 1) * C + kk] (nsid_downsample3_bwd_weight)
      const int C = q.K / 3;
      for (int v = 0; v < nseg; ++v) w.B[v] = static_cast<const char*>(q.x[v]) - (long)C * 2;
      w.ldb = 2 * C; w.pad_period = q.ds_out_nodes; w.pad_c1 = C;
    }
    w.slope = act_slope(q.act_in);
    long S = 1;                                  // splits per segment: rows / wgg_rows, whole 128-row multiples each
    if (cls != 2 && cls != 5 && cls != 6) {
      while (q.M % (2 * S) == 0 && (q.M / (2 * S)) % 128 == 0 && q.M / (2 * S) >= rows_cls) S *= 2;
      w.rchunk = (int)(q.M / S);
    } else {
      S = std::max<long>(1, (q.M + rows_cls - 1) / rows_cls);
      w.rchunk = (int)(((q.M + S - 1) / S + 31) / 32 * 32);
      S = (q.M + w.rchunk - 1) / w.rchunk;
    }
    w.seg_splits = (int)S;
    w.nsplit = (int)S * nseg;
    w.tiles = ((q.Nout + bm - 1) / bm) * ((q.K + bn - 1) / bn);
    const long tg = (long)w.tiles * q.groups;
    long nwg = tg * w.nsplit;
    if (w.nsplit < 8 && 8 % w.nsplit == 0) nwg = (tg + 8 / w.nsplit - 1) / (8 / w.nsplit) * 8;
    w.nwg = (int)nwg;
    NSID_REQUIRE(nwg < (1L << 30));
    if (ga[cls].n == WGG_MAXP || wgs[cls] + nwg > (1L << 30)) {
      const int rc = flush(cls);
      if (rc != NSID_OK) return rc;
    }
    ga[cls].wg0[ga[cls].n] = (int)wgs[cls];
    ga[cls].prob[ga[cls].n++] = w;
    wgs[cls] += (nwg + 7) / 8 * 8;
  }
  for (int c = 0; c < NCLS; ++c) {
    const int rc = flush(c);
    if (rc != NSID_OK) return rc;
  }
  return NSID_OK;
}

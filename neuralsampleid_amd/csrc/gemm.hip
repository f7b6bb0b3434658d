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
#include "nsid_common.h"

namespace {

inline float act_slope(int act) { return act == NSID_ACT_RELU ? 0.f : (act == NSID_ACT_LEAKY ? 0.2f : 1.f); }

struct GemmArgs {
  const float* A; long lda; long a_goff;   // group offset in elements
  const float* B; long ldb; long b_goff;
  float* C; long ldc; long c_goff;
  int I, J, R;
  // operand-load activation as a negative-side slope (1 = none, 0 = ReLU, 0.2 = LeakyReLU): one branch-free select
  const float* a_scale; const float* a_shift; float a_slope; long a_aff_goff;   // per-R affine on A (forward)
  const float* b_scale; const float* b_shift; float b_slope; long b_aff_goff;   // per-j affine on B (backward-wgt)
  const float* bias; long bias_goff;                                        // per-j
  const float* addend; long ldadd;                                          // same indexing as C
  float* stat; long stat_plane;    // stat[0*plane + tile*stat_ld + col], stat[1*plane + ...]
  long stat_ld;
  int rsplit;     // number of R-splits (grid.y); >1 => atomic epilogue
  int rchunk;     // R elements per split (multiple of 16)
  int atomic_out;
};

constexpr int BK = 16;

template <int ROWS, bool RMAJOR>
struct TileGeom {
  // RMAJOR: lds[row][BK+4] (row = i or j, R contiguous); else lds[BK][ROWS+4] (R = row, i/j contiguous)
  static constexpr int LD = RMAJOR ? (BK + 4) : (ROWS + 4);
  static constexpr int FLOATS = RMAJOR ? ROWS * LD : BK * LD;
  static constexpr int VEC = ROWS * BK / 4 / 256;   // float4 per thread per stage
};

// Staging is split in two so that the global-load latency hides under the MFMA block of the current stage
// (issue early / write late): stage_load only ISSUES the loads (raw operand quads plus the producer-BatchNorm
// scale/shift quads they will need); stage_store, which runs after the MFMAs, applies affine + activation, zero-fills
// out-of-range quads and writes LDS.  Nothing consumes a load result before the MFMA block.
template <int ROWS, bool RMAJOR>
struct StageRegs {
  static constexpr int VEC = TileGeom<ROWS, RMAJOR>::VEC;
  f32x4 v[VEC];
  f32x4 sc[RMAJOR ? VEC : 1], sh[RMAJOR ? VEC : 1];   // RMAJOR: affine follows the reduction index -> per stage
  bool ok[VEC];
};

template <int ROWS, bool RMAJOR>
__device__ __forceinline__ void stage_load(StageRegs<ROWS, RMAJOR>& s, const float* __restrict__ base, long ld,
                                           int row0, int nrows, int r0, int rend, const float* scale,
                                           const float* shift) {
  constexpr int VEC = TileGeom<ROWS, RMAJOR>::VEC;
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    const int idx = t + 256 * q;
    if (RMAJOR) {
      const int row = idx >> 2, rv = (idx & 3) * 4;   // 4 float4 per row of BK=16
      const int gi = row0 + row, gr = r0 + rv;
      s.ok[q] = gi < nrows && gr < rend;              // extents are multiples of 4: a quad is all-in or all-out
      // out-of-range quads read a valid dummy address (row/col 0 of the tile origin clamped) and are zeroed later
      const long off = s.ok[q] ? (long)gi * ld + gr : 0;
      s.v[q] = *reinterpret_cast<const f32x4*>(base + off);
      if (scale != nullptr) {
        const int ga = s.ok[q] ? gr : 0;
        s.sc[q] = *reinterpret_cast<const f32x4*>(scale + ga);
        s.sh[q] = *reinterpret_cast<const f32x4*>(shift + ga);
      }
    } else {
      constexpr int V_PER_ROW = ROWS / 4;
      const int rr = idx / V_PER_ROW, cv = (idx % V_PER_ROW) * 4;
      const int gr = r0 + rr, gc = row0 + cv;
      s.ok[q] = gr < rend && gc < nrows;
      const long off = s.ok[q] ? (long)gr * ld + gc : 0;
      s.v[q] = *reinterpret_cast<const f32x4*>(base + off);
    }
  }
}

// column-indexed affine of an i/j-major operand is the same for every stage: fetched once per kernel
template <int ROWS>
__device__ __forceinline__ void colaffine_load(f32x4* sc, f32x4* sh, int row0, int nrows, const float* scale,
                                               const float* shift) {
  constexpr int VEC = TileGeom<ROWS, false>::VEC;
  constexpr int V_PER_ROW = ROWS / 4;
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    const int cv = ((threadIdx.x + 256 * q) % V_PER_ROW) * 4;
    const int gc = row0 + cv < nrows ? row0 + cv : 0;
    sc[q] = *reinterpret_cast<const f32x4*>(scale + gc);
    sh[q] = *reinterpret_cast<const f32x4*>(shift + gc);
  }
}

template <int ROWS, bool RMAJOR>
__device__ __forceinline__ void stage_store(float* lds, const StageRegs<ROWS, RMAJOR>& s, bool affine, float slope,
                                            const f32x4* csc, const f32x4* csh) {
  constexpr int VEC = TileGeom<ROWS, RMAJOR>::VEC;
  constexpr int LD = TileGeom<ROWS, RMAJOR>::LD;
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    const int idx = t + 256 * q;
    f32x4 x = s.v[q];
    if (affine) {
      const f32x4 sc = RMAJOR ? s.sc[q] : csc[q];
      const f32x4 sh = RMAJOR ? s.sh[q] : csh[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = sc[e] * x[e] + sh[e];
        x[e] = v < 0.f ? v * slope : v;          // NaN compares false and passes through, as torch's activations do
      }
    }
    if (!s.ok[q]) x = f32x4{0.f, 0.f, 0.f, 0.f};
    if (RMAJOR) {
      const int row = idx >> 2, rv = (idx & 3) * 4;
      *reinterpret_cast<f32x4*>(lds + row * LD + rv) = x;
    } else {
      constexpr int V_PER_ROW = ROWS / 4;
      const int rr = idx / V_PER_ROW, cv = (idx % V_PER_ROW) * 4;
      *reinterpret_cast<f32x4*>(lds + rr * LD + cv) = x;
    }
  }
}

// fragment of one 16-row tile: element s feeds MFMA sub-step s (reduction index 4*(lane>>4)+s of the chunk)
template <int ROWS, bool RMAJOR>
__device__ __forceinline__ f32x4 frag_read(const float* lds, int row, int rq) {
  constexpr int LD = TileGeom<ROWS, RMAJOR>::LD;
  if (RMAJOR) {
    return *reinterpret_cast<const f32x4*>(lds + row * LD + 4 * rq);
  } else {
    f32x4 f;
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = lds[(4 * rq + s) * LD + row];
    return f;
  }
}

template <int BM, int BN, bool A_RMAJOR, bool B_RMAJOR>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
  using GA = TileGeom<BM, A_RMAJOR>;
  using GB = TileGeom<BN, B_RMAJOR>;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  constexpr int STAGE = GA::FLOATS + GB::FLOATS;
  constexpr int OUT_STAGE = 4 * 32 * (WN + 4);          // epilogue transpose buffers (4 waves x 32 rows)
  constexpr int LDS_FLOATS = 2 * STAGE > OUT_STAGE ? 2 * STAGE : OUT_STAGE;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tiles_j = (p.J + BN - 1) / BN;
  const int tiles_i = (p.I + BM - 1) / BM;
  int bid = blockIdx.x;
  {  // XCD-aware remap (blocks b and b+8 share an L2): give each XCD a contiguous run of tiles so the
     // column tiles that re-read one A row-panel hit the same L2. Bijective only when the grid divides by 8.
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
  }
  const int ti = bid / tiles_j, tj = bid % tiles_j;
  if (ti >= tiles_i) return;
  const int g = blockIdx.z;
  const int i0 = ti * BM, j0 = tj * BN;
  const int rbeg = blockIdx.y * p.rchunk;
  const int rend = min(p.R, rbeg + p.rchunk);

  const float* A = p.A + g * p.a_goff;
  const float* B = p.B + g * p.b_goff;
  const float* a_sc = p.a_scale ? p.a_scale + g * p.a_aff_goff : nullptr;
  const float* a_sh = p.a_shift ? p.a_shift + g * p.a_aff_goff : nullptr;
  const float* b_sc = p.b_scale ? p.b_scale + g * p.b_aff_goff : nullptr;
  const float* b_sh = p.b_shift ? p.b_shift + g * p.b_aff_goff : nullptr;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lr = lane & 15, rq = lane >> 4;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  StageRegs<BM, A_RMAJOR> ra;
  StageRegs<BN, B_RMAJOR> rb;
  // only these operand/affine pairings exist: forward (A reduction-major, affine on the reduction index) and
  // backward-weight (B column-major, affine on the column index)
  const bool a_aff = A_RMAJOR && a_sc != nullptr;
  const bool b_aff = !B_RMAJOR && b_sc != nullptr;
  f32x4 bcs[GB::VEC], bch[GB::VEC];
  if (b_aff) colaffine_load<BN>(bcs, bch, j0, p.J, b_sc, b_sh);
  const int nstage = (rend - rbeg + BK - 1) / BK;
  if (nstage > 0) {
    stage_load<BM, A_RMAJOR>(ra, A, p.lda, i0, p.I, rbeg, rend, a_aff ? a_sc : nullptr, a_sh);
    stage_load<BN, B_RMAJOR>(rb, B, p.ldb, j0, p.J, rbeg, rend, nullptr, nullptr);
    stage_store<BM, A_RMAJOR>(lds, ra, a_aff, p.a_slope, nullptr, nullptr);
    stage_store<BN, B_RMAJOR>(lds + GA::FLOATS, rb, b_aff, p.b_slope, bcs, bch);
  }
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const float* la = lds + (st & 1) * STAGE;
    const float* lb = la + GA::FLOATS;
    const bool more = st + 1 < nstage;
    if (more) {      // issue the next stage's global loads; they complete under the MFMA block below
      const int r0 = rbeg + (st + 1) * BK;
      stage_load<BM, A_RMAJOR>(ra, A, p.lda, i0, p.I, r0, rend, a_aff ? a_sc : nullptr, a_sh);
      stage_load<BN, B_RMAJOR>(rb, B, p.ldb, j0, p.J, r0, rend, nullptr, nullptr);
    }
    f32x4 fa[TM], fb[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) fa[a] = frag_read<BM, A_RMAJOR>(la, wm0 + 16 * a + lr, rq);
#pragma unroll
    for (int b = 0; b < TN; ++b) fb[b] = frag_read<BN, B_RMAJOR>(lb, wn0 + 16 * b + lr, rq);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
    if (more) {
      float* na = lds + ((st + 1) & 1) * STAGE;
      stage_store<BM, A_RMAJOR>(na, ra, a_aff, p.a_slope, nullptr, nullptr);
      stage_store<BN, B_RMAJOR>(na + GA::FLOATS, rb, b_aff, p.b_slope, bcs, bch);
    }
    __syncthreads();
  }

  // ---------------- epilogue. C/D layout: col = lane&15, row = 4*(lane>>4) + reg.
  float* C = p.C + g * p.c_goff;
  const float* bias = (p.bias && blockIdx.y == 0) ? p.bias + g * p.bias_goff : nullptr;
  float csum[TN], csq[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) csum[b] = csq[b] = 0.f;
  if (p.atomic_out) {
    // split-reduction outputs (weight gradients, split-K): one atomic per element
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int j = j0 + wn0 + 16 * b + lr;
      const bool jok = j < p.J;
      const float bj = (bias && jok) ? bias[j] : 0.f;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + wm0 + 16 * a + 4 * rq + r;
          if (i < p.I && jok) atomicAdd(C + (long)i * p.ldc + j, acc[a][b][r] + bj);
        }
    }
  } else {
    // Stores go through LDS so that every lane writes 16 contiguous bytes of an output row (4-byte stores from the
    // accumulator layout are store-issue bound). Each wave transposes its own 64 x WN sub-tile, 32 rows at a time.
    constexpr int OLD = WN + 4;                  // staggers rq groups over the banks, keeps rows 16-B aligned
    float* ost = lds + wave * (32 * OLD);        // 4 waves x 32 x (WN+4) floats <= one stage buffer
    constexpr int Q_PER_ROW = WN / 4, ROWS_PER_PASS = 64 / Q_PER_ROW;
    const int orow = lane / Q_PER_ROW, oq = (lane % Q_PER_ROW) * 4;
    const int jq = j0 + wn0 + oq;
    const bool jqok = jq < p.J;                  // J % 4 == 0: a quad is all-in or all-out
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    if (bias && jqok) bq = *reinterpret_cast<const f32x4*>(bias + jq);
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) ost[(16 * a2 + 4 * rq + r) * OLD + 16 * b + lr] = acc[2 * h + a2][b][r];
      __syncthreads();
#pragma unroll
      for (int pass = 0; pass < 32 / ROWS_PER_PASS; ++pass) {
        const int rr = pass * ROWS_PER_PASS + orow;
        const int i = i0 + wm0 + 32 * h + rr;
        if (i < p.I && jqok) {
          f32x4 v = *reinterpret_cast<const f32x4*>(ost + rr * OLD + oq);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bq[e];
          if (p.addend) {
            const f32x4 ad = *reinterpret_cast<const f32x4*>(p.addend + g * p.c_goff + (long)i * p.ldadd + jq);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += ad[e];
          }
          *reinterpret_cast<f32x4*>(C + (long)i * p.ldc + jq) = v;
        }
      }
      __syncthreads();
    }
    if (p.stat != nullptr) {
      // BatchNorm partial statistics of (acc + bias), from the accumulator registers
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int j = j0 + wn0 + 16 * b + lr;
        const bool jok = j < p.J;
        const float bj = (bias && jok) ? bias[j] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + wm0 + 16 * a + 4 * rq + r;
            if (i < p.I && jok) {
              const float v = acc[a][b][r] + bj;
              csum[b] += v;
              csq[b] += v * v;
            }
          }
      }
    }
  }
  if (p.stat != nullptr) {   // uniform branch; the stage buffers are free (loop ended on a barrier)
    float* red = lds;        // [2][2 wave-rows][BN]
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      float s = csum[b], q = csq[b];
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (rq == 0) {
        const int c = wn0 + 16 * b + lr;
        red[(0 * 2 + (wave >> 1)) * BN + c] = s;
        red[(1 * 2 + (wave >> 1)) * BN + c] = q;
      }
    }
    __syncthreads();
    if (threadIdx.x < BN) {
      const int j = j0 + threadIdx.x;
      if (j < p.J) {
        const long col = g * p.c_goff + j;     // forward: c_goff == Nout per group == column offset
        p.stat[(long)ti * p.stat_ld + col] = red[0 * BN + threadIdx.x] + red[1 * BN + threadIdx.x];
        p.stat[p.stat_plane + (long)ti * p.stat_ld + col] = red[2 * BN + threadIdx.x] + red[3 * BN + threadIdx.x];
      }
    }
  }
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

template <int BM, int BN, bool AR, bool BR>
int launch(const GemmArgs& p, int groups, hipStream_t s) {
  const int tiles = ((p.I + BM - 1) / BM) * ((p.J + BN - 1) / BN);
  dim3 grid(tiles, p.rsplit, groups);
  NSID_LAUNCH((gemm_kernel<BM, BN, AR, BR>), grid, dim3(256), 0, s, p);
  return nsid_launch_status();
}

}  // namespace

extern "C" int nsid_version(void) { return 1; }
extern "C" int nsid_row_tiles(int M) { return (M + NSID_ROW_TILE - 1) / NSID_ROW_TILE; }

extern "C" int nsid_linear_fwd(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo, int M,
                               int Nout, int K, int groups, const float* in_scale, const float* in_shift, int act_in,
                               int act_out, float* stat, int ksplit, void* stream) {
  NSID_REQUIRE(x && w && out && M > 0 && Nout > 0 && K > 0 && groups > 0 && ksplit >= 1);
  NSID_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && nsid_aligned16(x) && nsid_aligned16(w));
  NSID_REQUIRE(ldx >= groups * K && ldo >= groups * Nout);
  NSID_REQUIRE(Nout % 4 == 0 && ldo % 4 == 0 && nsid_aligned16(out) && (bias == nullptr || nsid_aligned16(bias)));
  NSID_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  NSID_REQUIRE(ksplit == 1 || (stat == nullptr && act_out == NSID_ACT_NONE));
  NSID_REQUIRE(act_in != NSID_ACT_ELU && (act_out == NSID_ACT_NONE || act_out == NSID_ACT_ELU));
  GemmArgs p{};
  p.A = x; p.lda = ldx; p.a_goff = K;
  p.B = w; p.ldb = K; p.b_goff = (long)Nout * K;
  p.C = out; p.ldc = ldo; p.c_goff = Nout;
  p.I = M; p.J = Nout; p.R = K;
  p.a_scale = in_scale; p.a_shift = in_shift; p.a_slope = act_slope(act_in); p.a_aff_goff = K;
  p.bias = bias; p.bias_goff = Nout;
  p.stat = stat; p.stat_ld = (long)groups * Nout; p.stat_plane = (long)nsid_row_tiles(M) * groups * Nout;
  p.rsplit = ksplit;
  p.rchunk = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
  p.atomic_out = ksplit > 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the statistics tile must be NSID_ROW_TILE rows, so BM = 128 always; narrow outputs take the 64-column tile
  const int rc = Nout <= 64 ? launch<128, 64, true, true>(p, groups, s) : launch<128, 128, true, true>(p, groups, s);
  if (rc != NSID_OK || act_out != NSID_ACT_ELU) return rc;
  const long n = (long)M * groups * Nout;
  NSID_LAUNCH(elu_inplace_kernel, dim3((int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, s, out,
              (long)M, groups * Nout, (long)ldo);
  return nsid_launch_status();
}

extern "C" int nsid_linear_bwd_data(const float* dout, int ldd, const float* w, const float* addend, int ldadd,
                                    float* din, int ldi, int M, int Nout, int K, int groups, void* stream) {
  NSID_REQUIRE(dout && w && din && M > 0 && Nout > 0 && K > 0 && groups > 0);
  NSID_REQUIRE(Nout % 4 == 0 && K % 4 == 0 && ldd % 4 == 0 && nsid_aligned16(dout) && nsid_aligned16(w));
  NSID_REQUIRE(ldd >= groups * Nout && ldi >= groups * K && ldi % 4 == 0 && nsid_aligned16(din));
  NSID_REQUIRE(addend == nullptr || (ldadd % 4 == 0 && ldadd >= groups * K && nsid_aligned16(addend)));
  GemmArgs p{};
  p.A = dout; p.lda = ldd; p.a_goff = Nout;
  p.B = w; p.ldb = K; p.b_goff = (long)Nout * K;     // B[R = n][j = k]
  p.C = din; p.ldc = ldi; p.c_goff = K;
  p.I = M; p.J = K; p.R = Nout;
  p.addend = addend; p.ldadd = ldadd;
  p.rsplit = 1; p.rchunk = (Nout + BK - 1) / BK * BK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (K <= 64) return launch<128, 64, true, false>(p, groups, s);
  return launch<128, 128, true, false>(p, groups, s);
}

extern "C" int nsid_linear_bwd_weight(const float* dout, int ldd, const float* x, int ldx, float* dw, int M, int Nout,
                                      int K, int groups, const float* in_scale, const float* in_shift, int act_in,
                                      void* stream) {
  NSID_REQUIRE(dout && x && dw && M > 0 && Nout > 0 && K > 0 && groups > 0);
  NSID_REQUIRE(Nout % 4 == 0 && K % 4 == 0 && ldd % 4 == 0 && ldx % 4 == 0 && nsid_aligned16(dout) && nsid_aligned16(x));
  NSID_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  GemmArgs p{};
  p.A = dout; p.lda = ldd; p.a_goff = Nout;          // A[R = m][i = n]
  p.B = x; p.ldb = ldx; p.b_goff = K;                // B[R = m][j = k]
  p.C = dw; p.ldc = K; p.c_goff = (long)Nout * K;
  p.I = Nout; p.J = K; p.R = M;
  p.b_scale = in_scale; p.b_shift = in_shift; p.b_slope = act_slope(act_in); p.b_aff_goff = K;
  p.atomic_out = 1;
  // 64x64 tiles when the output is narrow, or when 128x128 tiles times the deepest useful split (>= 512 rows each)
  // would leave most of the 256 CUs idle
  const long t128 = (long)((Nout + 127) / 128) * ((K + 127) / 128) * groups;
  const bool small = (Nout <= 64 || K <= 64) || t128 * ((M + 511) / 512) < 256;
  const int bm = small ? 64 : 128;
  const long tiles = (long)((Nout + bm - 1) / bm) * ((K + bm - 1) / bm) * groups;
  // split the row reduction so that about 2 workgroups per CU are in flight, each reducing >= 512 rows: every split
  // ends in a tile-sized burst of atomics (~1.3 TB/s chip-wide), so fewer, longer splits beat many short ones
  long want = (512 + tiles - 1) / tiles;
  long maxsplit = (M + 511) / 512;
  int rsplit = (int)(want < 1 ? 1 : (want > maxsplit ? maxsplit : want));
  p.rsplit = rsplit;
  p.rchunk = ((M + rsplit - 1) / rsplit + BK - 1) / BK * BK;
  p.rsplit = (M + p.rchunk - 1) / p.rchunk;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (small) return launch<64, 64, false, false>(p, groups, s);
  return launch<128, 128, false, false>(p, groups, s);
}

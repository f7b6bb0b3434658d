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

struct GemmArgs {
  const float* A; long lda; long a_goff;   // group offset in elements
  const float* B; long ldb; long b_goff;
  float* C; long ldc; long c_goff;
  int I, J, R;
  const float* a_scale; const float* a_shift; int a_act; long a_aff_goff;   // per-R affine on A (forward)
  const float* b_scale; const float* b_shift; int b_act; long b_aff_goff;   // per-j affine on B (backward-wgt)
  const float* bias; long bias_goff;                                        // per-j
  const float* addend; long ldadd;                                          // same indexing as C
  int out_act;
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

// global -> registers for one stage of one operand, with optional affine+activation, zero fill out of range
template <int ROWS, bool RMAJOR>
__device__ __forceinline__ void stage_load(f32x4* v, const float* __restrict__ base,
                                           long ld, int row0, int nrows, int r0, int rend, const float* scale,
                                           const float* shift, int act, bool affine_on_r) {
  constexpr int VEC = TileGeom<ROWS, RMAJOR>::VEC;
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    const int idx = t + 256 * q;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (RMAJOR) {
      const int row = idx >> 2, rv = (idx & 3) * 4;   // 4 float4 per row of BK=16
      const int gi = row0 + row, gr = r0 + rv;
      if (gi < nrows && gr < rend) {                  // R extents are multiples of 4, so a float4 is all-in or all-out
        x = *reinterpret_cast<const f32x4*>(base + (long)gi * ld + gr);
        if (scale != nullptr && affine_on_r) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + gr);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + gr);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = nsid_act(sc[e] * x[e] + sh[e], act);
        }
      }
    } else {
      constexpr int V_PER_ROW = ROWS / 4;
      const int rr = idx / V_PER_ROW, cv = (idx % V_PER_ROW) * 4;
      const int gr = r0 + rr, gc = row0 + cv;
      if (gr < rend && gc < nrows) {                  // column extents are multiples of 4 as well
        x = *reinterpret_cast<const f32x4*>(base + (long)gr * ld + gc);
        if (scale != nullptr && !affine_on_r) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + gc);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + gc);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = nsid_act(sc[e] * x[e] + sh[e], act);
        }
      }
    }
    v[q] = x;
  }
}

template <int ROWS, bool RMAJOR>
__device__ __forceinline__ void stage_store(float* lds, const f32x4* v) {
  constexpr int VEC = TileGeom<ROWS, RMAJOR>::VEC;
  constexpr int LD = TileGeom<ROWS, RMAJOR>::LD;
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    const int idx = t + 256 * q;
    if (RMAJOR) {
      const int row = idx >> 2, rv = (idx & 3) * 4;
      *reinterpret_cast<f32x4*>(lds + row * LD + rv) = v[q];
    } else {
      constexpr int V_PER_ROW = ROWS / 4;
      const int rr = idx / V_PER_ROW, cv = (idx % V_PER_ROW) * 4;
      *reinterpret_cast<f32x4*>(lds + rr * LD + cv) = v[q];
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
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

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

  f32x4 ra[GA::VEC], rb[GB::VEC];
  const int nstage = (rend - rbeg + BK - 1) / BK;
  if (nstage > 0) {
    stage_load<BM, A_RMAJOR>(ra, A, p.lda, i0, p.I, rbeg, rend, a_sc, a_sh, p.a_act, true);
    stage_load<BN, B_RMAJOR>(rb, B, p.ldb, j0, p.J, rbeg, rend, b_sc, b_sh, p.b_act, false);
    stage_store<BM, A_RMAJOR>(lds, ra);
    stage_store<BN, B_RMAJOR>(lds + GA::FLOATS, rb);
  }
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const float* la = lds + (st & 1) * STAGE;
    const float* lb = la + GA::FLOATS;
    const bool more = st + 1 < nstage;
    if (more) {
      const int r0 = rbeg + (st + 1) * BK;
      stage_load<BM, A_RMAJOR>(ra, A, p.lda, i0, p.I, r0, rend, a_sc, a_sh, p.a_act, true);
      stage_load<BN, B_RMAJOR>(rb, B, p.ldb, j0, p.J, r0, rend, b_sc, b_sh, p.b_act, false);
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
      stage_store<BM, A_RMAJOR>(na, ra);
      stage_store<BN, B_RMAJOR>(na + GA::FLOATS, rb);
    }
    __syncthreads();
  }

  // ---------------- epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + reg
  float* C = p.C + g * p.c_goff;
  const float* bias = (p.bias && blockIdx.y == 0) ? p.bias + g * p.bias_goff : nullptr;
  float csum[TN], csq[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) csum[b] = csq[b] = 0.f;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int j = j0 + wn0 + 16 * b + lr;
    const bool jok = j < p.J;
    const float bj = (bias && jok) ? bias[j] : 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wm0 + 16 * a + 4 * rq + r;
        if (i < p.I && jok) {
          float v = acc[a][b][r] + bj;
          csum[b] += v;
          csq[b] += v * v;
          if (p.addend) v += p.addend[g * p.c_goff + (long)i * p.ldadd + j];
          v = nsid_act(v, p.out_act);
          float* dst = C + (long)i * p.ldc + j;
          if (p.atomic_out) atomicAdd(dst, v);
          else *dst = v;
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
  NSID_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  NSID_REQUIRE(ksplit == 1 || (stat == nullptr && act_out == NSID_ACT_NONE));
  GemmArgs p{};
  p.A = x; p.lda = ldx; p.a_goff = K;
  p.B = w; p.ldb = K; p.b_goff = (long)Nout * K;
  p.C = out; p.ldc = ldo; p.c_goff = Nout;
  p.I = M; p.J = Nout; p.R = K;
  p.a_scale = in_scale; p.a_shift = in_shift; p.a_act = act_in; p.a_aff_goff = K;
  p.bias = bias; p.bias_goff = Nout;
  p.out_act = act_out;
  p.stat = stat; p.stat_ld = (long)groups * Nout; p.stat_plane = (long)nsid_row_tiles(M) * groups * Nout;
  p.rsplit = ksplit;
  p.rchunk = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
  p.atomic_out = ksplit > 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the statistics tile must be NSID_ROW_TILE rows, so BM = 128 always; narrow outputs take the 64-column tile
  if (Nout <= 64) return launch<128, 64, true, true>(p, groups, s);
  return launch<128, 128, true, true>(p, groups, s);
}

extern "C" int nsid_linear_bwd_data(const float* dout, int ldd, const float* w, const float* addend, int ldadd,
                                    float* din, int ldi, int M, int Nout, int K, int groups, void* stream) {
  NSID_REQUIRE(dout && w && din && M > 0 && Nout > 0 && K > 0 && groups > 0);
  NSID_REQUIRE(Nout % 4 == 0 && K % 4 == 0 && ldd % 4 == 0 && nsid_aligned16(dout) && nsid_aligned16(w));
  NSID_REQUIRE(ldd >= groups * Nout && ldi >= groups * K);
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
  p.b_scale = in_scale; p.b_shift = in_shift; p.b_act = act_in; p.b_aff_goff = K;
  p.atomic_out = 1;
  const bool small = (Nout <= 64 || K <= 64);
  const int bm = small ? 64 : 128;
  const long tiles = (long)((Nout + bm - 1) / bm) * ((K + bm - 1) / bm) * groups;
  // split the row reduction so that about 1024 workgroups are in flight, each reducing >= 256 rows
  long want = (1024 + tiles - 1) / tiles;
  long maxsplit = (M + 255) / 256;
  int rsplit = (int)(want < 1 ? 1 : (want > maxsplit ? maxsplit : want));
  p.rsplit = rsplit;
  p.rchunk = ((M + rsplit - 1) / rsplit + BK - 1) / BK * BK;
  p.rsplit = (M + p.rchunk - 1) / p.rchunk;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (small) return launch<64, 64, false, false>(p, groups, s);
  return launch<128, 128, false, false>(p, groups, s);
}

/* nsid.h — C ABI of libnsid_hip.so: the MI355X (gfx950) kernels behind the GNN contrastive-fingerprint path.
 *
 * The reference (chymaera96/NeuralSampleID) has no FFI: its boundary is Python nn.Module.forward.  Each entry
 * point below replaces the chain of stock ATen ops that one reference function issues (file:line cited per
 * entry, paths relative to the reference repo); INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - every pointer is DEVICE memory, fp32 unless typed otherwise, 16-byte aligned, row-major;
 *  - features are node-major rows: a (B, C, N, 1) reference tensor is the matrix X[B*N][C] (row = b*N + n);
 *  - `stream` is a hipStream_t; calls only enqueue work: no allocation, no synchronisation, no retained
 *    pointers, safe under hipGraph stream capture;
 *  - return value: NSID_OK, or a negative NSID_E* code (the Python host raises RuntimeError on it);
 *  - buffers documented "+=" are accumulated into (zero them first for a fresh gradient).
 */
#ifndef NSID_H
#define NSID_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSID_OK 0
#define NSID_EINVAL (-1)   /* unsupported shape / misaligned pointer / bad argument */
#define NSID_ELAUNCH (-2)  /* the HIP runtime refused the launch */

#define NSID_ACT_NONE 0
#define NSID_ACT_RELU 1
#define NSID_ACT_LEAKY 2 /* LeakyReLU(0.2), encoder/graph_encoder.py:153 */
#define NSID_ACT_ELU 3   /* nn.ELU, simclr/simclr.py:26 */

#define NSID_ROW_TILE 128 /* rows per BatchNorm partial-statistics tile (all kernels agree on it) */

/* storage type of ACTIVATION tensors (features, raw conv outputs, their gradients): every `void*` activation argument
 * below comes with a dtype code. Parameters, parameter gradients, BatchNorm vectors/statistics, indices and the small
 * head tensors (pooled features, projector, embeddings) are always fp32. Arithmetic is always fp32-accumulated.
 * bf16 rows need C % 8 == 0 (16-byte chunks). */
#define NSID_F32 0
#define NSID_BF16 1

#define NSID_GEMM_FP32 0 /* fp32 operands on v_mfma_f32_16x16x4_f32: exact fp32, the parity path (default) */
#define NSID_GEMM_BF16 1 /* operands rounded to bf16 while staged into LDS, fp32 storage + fp32 accumulate */

int nsid_version(void);
/* debug: install (or clear with NULL) a device buffer of 4 x uint64 per workgroup; the GEMM kernels then record
   {start, end of main loop, end} on the 100 MHz clock and (XCC_ID << 32 | HW_ID). tools/gemm_trace.py reads it. */
int nsid_debug_gemm_trace(void* device_buf);
/* same for the kNN kernel: {start, features staged, normalised, end} per workgroup (= clip) */
int nsid_debug_knn_trace(void* device_buf);
/* process-wide arithmetic of the nsid_linear_* GEMMs (BASELINE config 2 names bf16 compute); returns NSID_OK/EINVAL */
int nsid_set_gemm_precision(int mode);
int nsid_get_gemm_precision(void);
/* ---- tuning: named launch-heuristic constants (tile-width thresholds, workgroup caps, split targets, kernel-family switches).
   The library never reads the environment: defaults are compiled in (csrc/nsid_common.h NSID_TUNING_TABLE lists every key with
   its default and meaning), and a run changes them only through these calls, so kernel selection and arithmetic depend on explicit
   caller state alone. Keys used by the tests: "g256_min", "g256_train", "ffn256", "knn_strips". Unknown key -> NSID_EINVAL. */
int nsid_set_tuning(const char* key, long value);
int nsid_get_tuning(const char* key, long* value);
int nsid_reset_tuning(void);                 /* every key back to its compiled-in default */
int nsid_tuning_count(void);
const char* nsid_tuning_key(int i);          /* i in [0, nsid_tuning_count()) */
/* debug: launch counters per kernel VARIANT since the last reset (csrc/nsid_common.h NSID_COUNTER_TABLE lists the keys, e.g.
   "gemm_full", "gemm_ks2", "wgrad_rect", "wgrad3", "gemm_bn_sums", "knn2"); -1 for an unknown key. Host-side: counted when enqueued. */
long nsid_debug_counter(const char* key);
int nsid_debug_counters_reset(void);
int nsid_debug_counter_count(void);
const char* nsid_debug_counter_key(int i);
/* launches of nsid_linear_fwd / nsid_linear_fwd_res that took the 256x256-tile LDS-DMA kernel (csrc/gemm256.hip) so far */
long nsid_gemm_g256_launches(void);
/* number of NSID_ROW_TILE row tiles of an M-row matrix: size of the partial-statistics buffers */
int nsid_row_tiles(int M);
/* Bytes of caller-provided scratch an op needs, or -1 for an unknown op name ("a nsid_workspace_bytes(op, dims)
 * query per op"; the reference has no counterpart — its ops allocate through torch). No entry point allocates; the kernels keep
 * their working sets in LDS and registers, so every op answers 0 ("knn_graph", "mr_aggregate", "linear", "linear_bwd_data",
 * "linear_bwd_weight", "downsample3", "peak_patchify", "bn_apply", "node_mean", "l2norm", "adam", "ffn_fused", "mrconv_fused")
 * except: "bn_stat" (rows x cols layer: the [2][nsid_row_tiles(rows)][cols] fp32 partial sums between a GEMM's statistics
 * epilogue / nsid_bn_bwd_reduce and the finalize kernels), "ntxent" (rows = pairs of the global batch: nsid_ntxent_ws_floats),
 * "sumsq" (rows = gradient elements: nsid_sumsq_blocks partial sums). */
long nsid_workspace_bytes(const char* op, long rows, long cols);

/* ---- 1x1 convolution / Linear as a row GEMM on MFMA (fp32 accumulate) ------------------------------------
 * act_dtype = NSID_BF16: the activation operands/outputs are bf16 in HBM (bias, statistics, weight gradients stay
 * fp32) and the bf16 MFMA path is used whatever nsid_set_gemm_precision says.  w_dtype = NSID_BF16 (only together
 * with act_dtype = NSID_BF16): `w` points to a bf16 copy of the weight matrix (nsid_f32_to_bf16; the optimiser keeps
 * one shadow buffer for all parameters) — same values the fp32 weights round to when staged, half the operand bytes.
 * Replaces nn.Conv2d(…,1) / nn.Linear forward+backward at encoder/gcn_lib/torch_vertex.py:152-162,
 * encoder/graph_encoder.py:74-77,151,179, encoder/gcn_lib/torch_nn.py:56 (groups=4), simclr/simclr.py:25-28.
 *
 * forward:  out[m, g*Nout+n] = act_out( bias[g*Nout+n] + sum_k f(x[m, g*K+k]) * w[g*Nout+n, k] )
 *           f(v) = act_in(in_scale[g*K+k]*v + in_shift[g*K+k])  — the producer's BatchNorm(+activation) applied
 *           on load, so normalised activations are never materialised (in_scale == NULL: f = identity).
 *           stat (optional): [2][nsid_row_tiles(M)][groups*Nout] per-row-tile column sums / sums of squares of the
 *           pre-activation output, the input of nsid_bn_finalize (training-mode BatchNorm statistics).
 *           ksplit > 1 splits K over workgroups and accumulates atomically: `out` must be zeroed, stat == NULL,
 *           act_out == NSID_ACT_NONE.  act_out: NSID_ACT_NONE, NSID_ACT_ELU (fp32 storage only) or NSID_ACT_RELU (stat == NULL,
 *           ksplit == 1: eval mode with the BatchNorm folded into (w, bias) writes relu(BN(conv)) and its consumer loads plain values). */
int nsid_linear_fwd(const void* x, int ldx, const void* w, int w_dtype, const float* bias, void* out, int ldo, int M,
                    int Nout, int K, int groups, const float* in_scale, const float* in_shift, int act_in, int act_out,
                    float* stat, int ksplit, int act_dtype /* of x and out */, void* stream);
/* the same GEMM with `addend` (same storage type and row layout as out, row stride ldadd) added in the epilogue:
   out = f(x) W^T + bias + addend. One launch for "conv + eval-mode BatchNorm folded into (W, bias) + shortcut"
   (reference: x = self.fc2(x) ... + shortcut, torch_vertex.py:183-195, graph_encoder.py:82-89, in eval mode). */
int nsid_linear_fwd_res(const void* x, int ldx, const void* w, int w_dtype, const float* bias, const void* addend, int ldadd,
                        void* out, int ldo, int M, int Nout, int K, int groups, const float* in_scale,
                        const float* in_shift, int act_in, int act_dtype, void* stream);
/* eval-mode FFN in ONE launch: out = x + W2 relu(W1 x + b1) + b2 with both BatchNorms folded into (W1, b1) (H x C) and (W2, b2)
   (C x H) — FFN.forward, encoder/graph_encoder.py:82-89, in eval mode. x, out: M x C bf16 contiguous; W1, W2: bf16 row-major;
   b1, b2 fp32. The M x H hidden tensor never reaches HBM. Returns 1 (nothing launched) outside C in {64, 128} with M % 128 == 0 or
   C = 256 with M % 256 == 0 (csrc/ffn256_fused.hip; tuning key "ffn256"), H = 4C: the caller then runs nsid_linear_fwd +
   nsid_linear_fwd_res. */
int nsid_ffn_fused_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int M, int C,
                       int H, void* stream);
/* eval-mode MRConv2d in ONE launch, one workgroup per clip: v = relu(W (*)_4 [y, max_j(y[idx_j] - y)] + b) with the BatchNorm folded
   into (W, b) — MRConv2d.forward (gcn_lib/torch_vertex.py:19-34) + BasicConv (torch_nn.py:52-76) in eval mode. y: (B*N, C) bf16 plain
   values (the producer's BatchNorm folded too), idx: (B, N, k) clip-local, w: (2C, C/2) bf16, bias: (2C) fp32, out: (B*N, 2C) bf16.
   The interleaved (B*N, 2C) tensor of nsid_mr_aggregate_fwd is never formed. Returns 1 (nothing launched) outside C in {64, 128, 256},
   N*C = 16384, k <= 64: the caller then runs nsid_mr_aggregate_fwd + nsid_linear_fwd. */
int nsid_mrconv_fused_fwd(const void* y, const int32_t* idx, int B, int N, int C, int k, const void* w, const float* bias, void* out,
                          void* stream);
/* backward-data: din[m, g*K+k] = addend[m, g*K+k] + sum_n dout[m, g*Nout+n] * w[g*Nout+n, k]   (addend optional) */
int nsid_linear_bwd_data(const void* dout, int ldd, const void* w, int w_dtype, const void* addend, int ldadd,
                         void* din, int ldi, int M, int Nout, int K, int groups,
                         int act_dtype /* dout, addend, din */, void* stream);
/* backward-data whose output din IS dL/dy of a BatchNorm(+activation) layer y = act(BN(r)), r (M x groups*K, bf16, same
 * layout as din) being that layer's raw input: additionally writes bn_partial[2][nsid_row_tiles(M)][groups*K], the
 * per-row-tile column sums of g = din*act'(scale*r+shift) and g*xhat — exactly what nsid_bn_bwd_reduce(din, r, ...) would
 * produce from the stored din (one launch and two tensor reads less per layer). bf16 storage only, ldi == groups*K. */
int nsid_linear_bwd_data_bn(const void* dout, int ldd, const void* w, int w_dtype, const void* addend, int ldadd,
                            void* din, int ldi, int M, int Nout, int K, int groups, int act_dtype, const void* bn_r,
                            const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                            int bn_act, float* bn_partial, void* stream);
/* backward-data of a conv+BatchNorm(+act) layer whose BatchNorm backward is evaluated ON THE OPERAND LOAD: with dy = dL/d act(BN(r))
 * and r that layer's raw conv output (both M x groups*Nout, bf16, contiguous), dr = BN-backward(dy, r) = sc*g + P*r + Q,
 * g = dy*act'(sc*r + sh), coef4[4][groups*Nout] = {sc, sh, P, Q} from nsid_bn_bwd_finalize_fused, the call computes
 * din = addend + dr w  and writes dr (bf16, M x groups*Nout) once for nsid_linear_bwd_weight. It replaces the
 * nsid_bn_bwd_apply pass + nsid_linear_bwd_data[_bn] pair of a Conv2d+BatchNorm2d(+ReLU) site (encoder/graph_encoder.py:74-77,
 * gcn_lib/torch_nn.py:56-60, torch_vertex.py:152-155). bn_* as in nsid_linear_bwd_data_bn (bn_r NULL: none).
 * dr must not alias dy or r (NSID_EINVAL): only column tile 0 of a row panel writes dr while the other column tiles read dy and r.
 * Returns 1, having launched nothing, when the shape is outside the fused form (caller: run the two-call form). */
int nsid_linear_bwd_data_bnapply(const void* dy, const void* r, const float* coef4, int act, void* dr, const void* w, int w_dtype,
                                 const void* addend, int ldadd, void* din, int ldi, int M, int Nout, int K, int groups,
                                 int act_dtype, const void* bn_r, const float* bn_scale, const float* bn_shift,
                                 const float* bn_mean, const float* bn_invstd, int bn_act, float* bn_partial, void* stream);
/* backward-weight: dw[g*Nout+n, k] += sum_m dout[m, g*Nout+n] * f(x[m, g*K+k])   (f as in forward; atomic) */
int nsid_linear_bwd_weight(const void* dout, int ldd, const void* x, int ldx, float* dw, int M, int Nout, int K,
                           int groups, const float* in_scale, const float* in_shift, int act_in,
                           int act_dtype /* dout and x */, void* stream);
/* Many weight gradients in ONE launch per tile class: the deferred weight-gradient phase of a training step (nothing reads a weight
 * gradient before train.py:73-75's clip + step, so the conv layers of encoder/gcn_lib/torch_vertex.py:152-162,
 * encoder/graph_encoder.py:74-77 and gcn_lib/torch_nn.py:56 record their problem during backward and issue all of them here).
 * Problem i: dw += sum over its row segments v of dout[v]^T f_v(x[v]) with f_v = act_in(in_scale[v] * x + in_shift[v]) (in_scale[v] NULL:
 * identity; both segments with or without). The two segments are the two views of a contrastive step: the same layer, M rows each,
 * the same leading dimensions; dout[1] = x[1] = NULL: one segment. act_dtype: the storage of every dout / x of the call (fp32: the
 * projector head's tensors; the tile classes with 64-deep stages exist for bf16 only). The problem table is
 * copied into the kernel arguments: the array may be freed as soon as the call returns, the launch is capturable. */
typedef struct nsid_wgrad_problem {
  const void* dout[2];
  const void* x[2];
  const float* in_scale[2];
  const float* in_shift[2];
  float* dw;
  int ldd, ldx, M, Nout, K, groups, act_in;
  int ds_out_nodes;   /* 0: a plain row GEMM. > 0: the packed weight gradient of a Downsample (Conv2d 3x3 s2 p1 on a width-1 map,
                         encoder/graph_encoder.py:44): x[v] is its (B*N, K/3) input, read as the zero-padded 3-tap view of
                         nsid_downsample3_bwd_weight, dout[v] its (M = B*N/2, Nout) output gradient, dw the packed (Nout, K) gradient,
                         ds_out_nodes = N/2; ldx = K/3, groups = 1, no affine */
} nsid_wgrad_problem;
int nsid_linear_bwd_weight_grouped(const nsid_wgrad_problem* problems, int n, int act_dtype,
                                   int max_workgroups /* 0: one workgroup per work item; > 0: at most this many, each walking several
                                                         items: a launch that runs beside other kernels of the step */,
                                   void* stream);
/* out[c] += sum_m x[m, c]  (bias gradients) */
int nsid_colsum_acc(const void* x, int ldx, int M, int C, float* out, int dtype, void* stream);

/* ---- BatchNorm2d, training mode, split around the GEMMs ------------------------------------------------
 * Replaces nn.BatchNorm2d at encoder/graph_encoder.py:45,75,77,152, torch_vertex.py:154,161, torch_nn.py:32.
 * finalize: reduces the GEMM's partial statistics (fp64), writes scale = gamma*invstd, shift = beta-mean*scale,
 * mean, invstd, and updates running_mean / running_var (unbiased) / num_batches_tracked (momentum 0.1). */
int nsid_bn_finalize(const float* stat, int tiles, int C, int M, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                     float eps, float* scale, float* shift, float* mean, float* invstd, void* stream);
/* Training-mode finalize WITHOUT the running-statistics update, plus the unbiased variance (uvar) that update needs; and
   the deferred update itself for n layers in one launch per 16 layers: running = (1-m)*running + m*stat for view a, then
   (when mean_b / uvar_b are given) for view b — the order the reference applies them (simclr/simclr.py:36,42). The pointer
   arrays live in host memory. */
int nsid_bn_finalize_deferred(const float* stat, int tiles, int C, int M, const float* gamma, const float* beta, float eps,
                              float* scale, float* shift, float* mean, float* invstd, float* uvar, void* stream);
int nsid_bn_running_update(int n, const int* C, float* const* running_mean, float* const* running_var,
                           int64_t* const* num_batches_tracked, const float* const* mean_a, const float* const* uvar_a,
                           const float* const* mean_b, const float* const* uvar_b, float momentum, void* stream);
/* eval mode: scale/shift from the running statistics */
int nsid_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, void* stream);
/* out = act(scale*r + shift) + residual   (residual optional; materialises the residual stream) */
int nsid_bn_apply(const void* r, const float* scale, const float* shift, int act, const void* residual,
                  void* out, int M, int C, int dtype, void* stream);
/* backward, step 1: g = dout * act'(scale*r+shift); partial[2][tiles][C] = per-tile sums of g and g*xhat */
int nsid_bn_bwd_reduce(const void* dout, const void* r, int M, int C, const float* scale, const float* shift,
                       const float* mean, const float* invstd, int act, float* partial, int dtype, void* stream);
/* step 2: dgamma += sum g*xhat; dbeta += sum g; coef[2][C] = {sum g / M, sum g*xhat / M}; tiles = rows of partial sums (any count) */
int nsid_bn_bwd_finalize(const float* partial, int tiles, int C, int M, float* dgamma, float* dbeta, float* coef,
                         void* stream);
/* step 2 for a consumer that evaluates step 3 on its operand load (nsid_linear_bwd_data_bnapply): additionally
   coef4[4][C] = {scale, shift, P, Q} with dr = scale*g + P*r + Q, P = -scale*coef1*invstd, Q = scale*(coef1*invstd*mean - coef0) */
int nsid_bn_bwd_finalize_fused(const float* partial, int tiles, int C, int M, float* dgamma, float* dbeta, float* coef,
                               const float* scale, const float* shift, const float* mean, const float* invstd, float* coef4,
                               void* stream);
/* step 3: dr = scale * (g - coef0 - xhat*coef1)   (dr may alias dout) */
int nsid_bn_bwd_apply(const void* dout, const void* r, int M, int C, const float* scale, const float* shift,
                      const float* mean, const float* invstd, int act, const float* coef, void* dr, int dtype,
                      void* stream);

/* ---- dilated kNN graph ---------------------------------------------------------------------------------
 * Replaces DenseDilatedKnnGraph.forward = F.normalize + pairwise_distance + topk + [::dilation]
 * (encoder/gcn_lib/torch_edge.py:270-284, 70-103, 7-18, 245-255).  y = scale*r+shift (scale NULL: y = r) is
 * L2-normalised over channels, D = |a|^2 - 2ab + |b|^2 is formed per clip in LDS (never written to HBM),
 * the k*dilation nearest are selected in ascending distance (ties: lower index first) and every dilation-th is
 * kept.  idx[(b*N+n)*k + j] is clip-local (0..N-1), int32; the reference's edge_index[1] (centre) is implicit.
 * Limits (checked, NSID_EINVAL otherwise): N % 32 == 0, C % 16 == 0, ldr >= C, k*dilation <= N, r 16-byte
 * aligned; ldr % 4 == 0 (fp32) / % 8 == 0 (bf16). Graphs of more than 256 nodes, or clips of more than 32 768 features
 * (encoder/graph_encoder.py:144 with a cfg beyond grafp.yaml's 64 x 128 input), take a form with one workgroup per
 * 16-row strip (same arithmetic, features re-read from L2): 16 (C + 4) + 19 N + 64 floats of LDS must fit 160 KB
 * (N <= ~2 000). */
int nsid_knn_graph(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C, int k,
                   int dilation, int32_t* idx, int dtype, void* stream);

/* ---- max-relative aggregation --------------------------------------------------------------------------
 * Replaces MRConv2d.forward up to the grouped conv: 2x batched_index_select, max_k(x_j - x_i), interleave
 * (encoder/gcn_lib/torch_vertex.py:21-32, torch_nn.py:79-98).
 * u[row, 2c] = y[row, c]; u[row, 2c+1] = max_j y[b*N+idx[row,j], c] - y[row, c];  argmax[row, c] = winning j
 * (first maximum, as torch.max). backward routes du to the winning neighbour and the centre. k <= 255. */
int nsid_mr_aggregate_fwd(const void* r, int ldr, const float* scale, const float* shift, const int32_t* idx, int B,
                          int N, int C, int k, void* u, uint8_t* argmax, int dtype, void* stream);
int nsid_mr_aggregate_bwd(const void* du, const int32_t* idx, const uint8_t* argmax, int B, int N, int C, int k,
                          void* dy, int dtype, void* stream);
/* backward of the aggregation when its input was y = act(BN(r)) of a conv+BatchNorm layer (Grapher fc1, encoder/gcn_lib/torch_vertex.py:
 * 183-195): dy as above, plus step 1 of that BatchNorm's backward over the clip's rows, partial[2][B][C] (one row per clip:
 * nsid_bn_bwd_finalize[_fused] takes tiles = B), from the ROUNDED dy -- what nsid_bn_bwd_reduce would compute from the stored dy.
 * bf16 storage, N*C = 16384, C a power of two in [64, 512] (the encoder's clip at every stage); NSID_EINVAL otherwise. */
int nsid_mr_aggregate_bwd_bn(const void* du, const int32_t* idx, const uint8_t* argmax, int B, int N, int C, int k, void* dy,
                             const void* bn_r, int bn_ldr, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                             const float* bn_invstd, int bn_act, float* partial, int dtype, void* stream);

/* batched_index_select(x, idx) of the reference (encoder/gcn_lib/torch_nn.py:79-98) in the reference's own layouts:
 * x (B, C, N) fp32, idx (B, Nq, k) int32 clip-local -> out (B, C, Nq, k), out[b,c,n,j] = x[b,c,idx[b,n,j]].
 * The hot path never materialises this tensor (nsid_mr_aggregate_fwd gathers while it aggregates); the entry serves
 * drop-in callers of the Python symbol. bwd: dx[b,c,idx[b,n,j]] += dout[b,c,n,j] (zero dx first). */
int nsid_batched_index_select_fwd(const float* x, const int32_t* idx, int B, int C, int N, int Nq, int k, float* out,
                                  void* stream);
int nsid_batched_index_select_bwd(const float* dout, const int32_t* idx, int B, int C, int N, int Nq, int k,
                                  float* dx /* += */, void* stream);

/* ---- Downsample: Conv2d 3x3 stride 2 pad 1 on a width-1 map (encoder/graph_encoder.py:44) ---------------
 * Only kernel column 1 meets data, so it is a 3-tap stride-2 conv along N = one GEMM over gathered rows:
 * col[b*No+n', t*C+c] = x[b*N + 2n'-1+t, c] (0 outside), wp[o, t*C+c] = w[o, c, t, 1]. */
/* Without im2col (N even): the im2col matrix is a zero-padded strided VIEW of x (row stride 2C, K = 3C, base x - C; the
 * first C columns of the first output node of every clip are the left padding), read in place by the GEMM's operand loads.
 *   fwd        : out[B*No][Cout] = col . wp^T + bias (+ BatchNorm partial statistics `stat`, may be NULL); wp as below
 *   bwd_weight : dwp[Cout][3C] += dout^T . col   (then nsid_unpack_ds_wgrad)
 *   bwd_data   : dx[2n'] = dout[n'] . W_1,  dx[2n'+1] = dout[n'] . W_2 + dout[n'+1] . W_0  (two GEMMs, no col2im);
 *                w_odd = [W_2 ; W_0] as (2*Cout, C) from nsid_pack_ds_weight_bwd; wp / w_odd share w_dtype. */
int nsid_downsample3_fwd(const void* x, int B, int N, int C, const void* wp, int w_dtype, const float* bias, void* out,
                         int Cout, float* stat, int act_dtype, void* stream);
int nsid_downsample3_bwd_weight(const void* dout, const void* x, float* dwp /* += */, int B, int N, int C, int Cout,
                                int act_dtype, void* stream);
int nsid_downsample3_bwd_data(const void* dout, const void* wp, const void* w_odd, int w_dtype, void* dx, int B, int N,
                              int C, int Cout, int act_dtype, void* stream);
int nsid_pack_ds_weight_bwd(const float* w, int Cout, int Cin, float* w_odd, void* stream);
/* The materialising form (any N): */
int nsid_im2col3_fwd(const void* x, int B, int N, int C, void* col, int dtype, void* stream);
int nsid_im2col3_bwd(const void* dcol, int B, int N, int C, void* dx, int dtype, void* stream);
int nsid_pack_ds_weight(const float* w, int Cout, int Cin, float* wp, void* stream);
/* All Downsample layers of a model at once (n <= 8; the pointer arrays live in host memory), once per training step instead of per layer
 * and view: nsid_ds_prepack writes the packed forward weight (Cout, 3*Cin) and the packed backward weight [W_2 ; W_0] (2*Cout, Cin) of
 * every layer straight to bf16 and zeroes its packed gradient buffers dwp (TWO of them, contiguous: (2, Cout, 3*Cin) fp32, one per view of a
 * contrastive step; dwp or its entries may be NULL). Each view unpacks its own packed gradient with nsid_unpack_ds_wgrad (atomic adds). */
int nsid_ds_prepack(int n, const float* const* w, void* const* wp16, void* const* wb16, float* const* dwp, const int* Cout,
                    const int* Cin, void* stream);
int nsid_unpack_ds_wgrad(const float* dwp, int Cout, int Cin, float* dw /* += */, void* stream);

/* ---- GPUPeakExtractorv2 (peak_extractor.py:45-70) ------------------------------------------------------
 * per-clip min-max normalise, [time ramp, freq ramp, spec] -> Conv2d(3->F, kernel=stride=(pb,pf)) + ReLU.
 * out is node-major [B*(H/pb)*(W/pf)][ldo] (columns >= F untouched); minmax[B][2] is kept for backward.
 * backward gives the conv weight/bias gradients only (the spectrogram is data).
 * A clip is staged whole in LDS by the forward (H (W + 8) floats <= ~155 KB: 64 x 128 and 256 x 128 inputs both fit); the backward walks a
 * clip that exceeds 64 KB in bands of patch rows. */
int nsid_peak_patchify_fwd(const float* spec, const float* w, const float* bias, int B, int H, int W, int pb, int pf,
                           int F, void* out, int ldo, float* minmax, int out_dtype, void* stream);
int nsid_peak_patchify_bwd(const float* spec, const float* minmax, const void* out, const void* dout, int ldo,
                           int B, int H, int W, int pb, int pf, int F, float* dw /* += */, float* dbias /* += */,
                           int out_dtype, void* stream);
/* the same with a workspace of B * 776 floats for GraFP's patch (pb = 4, pf = 8, F = 8, 64 x 128 clips; NSID_EINVAL otherwise):
 * per-clip partial sums by plain stores + one reduce launch instead of B * 776 atomics onto 776 addresses */
int nsid_peak_patchify_bwd_ws(const float* spec, const float* minmax, const void* out, const void* dout, int ldo, int B, int H,
                              int W, int pb, int pf, int F, float* dw, float* dbias, float* ws, int out_dtype, void* stream);

/* ---- node mean (encoder/graph_encoder.py:211), ELU', L2 normalise (simclr/simclr.py:38,44) --------------*/
int nsid_node_mean_fwd(const void* x, int B, int N, int C, float* out, int x_dtype, void* stream);
int nsid_node_mean_bwd(const float* dout, int B, int N, int C, void* dx, int dx_dtype, void* stream);
int nsid_elu_bwd(const float* dout, const float* out, long n, float* din, void* stream);
int nsid_l2norm_fwd(const float* p, int B, int d, float eps, float* z, float* norm, void* stream);
int nsid_l2norm_bwd(const float* dz, const float* z, const float* norm, int B, int d, float eps, float* dp,
                    void* stream);

/* ---- log-mel front end (modules/transformations.py:27-34 MelSpectrogram + AmplitudeToDB, :94-105 unfold) -------
 * waveform -> reflect pad (n_fft/2 each side, torch.stft center=True) -> STFT as a fp32 GEMM through nsid_linear_fwd
 * (x = padded waveform with ldx = hop: overlapping frames; w = [window*cos ; -window*sin] rows, (2*n_freq) x n_fft) ->
 * power, HTK-mel filterbank (fb dense [n_mels][n_freq], band[m] = [first, last+1) non-zero bins), 10*log10(max(.,1e-10))
 * -> logmel (n_mels, T) -> segments (S, n_mels, n_frames) with hop `step` frames. */
int nsid_reflect_pad(const float* x, long L, int pad, float* out, void* stream);
int nsid_power_mel_db(const float* spec, long ld, int n_freq, const float* fb, const int* band, int n_mels, int T,
                      float* out, void* stream);
int nsid_unfold_segments(const float* logmel, int n_mels, int T, int n_frames, int step, int S, float* out, void* stream);

/* bf16 shadow of fp32 weights: dst[i] = bf16_rne(src[i]), n % 8 == 0, both 16-byte aligned (operand `w` of
 * nsid_linear_fwd / nsid_linear_bwd_data with w_dtype = NSID_BF16). */
int nsid_f32_to_bf16(const float* src, void* dst, long n, void* stream);

/* ---- NT-Xent (simclr/ntxent.py:5-30) -------------------------------------------------------------------
 * Row i = 2p+v of the interleaved (2*Bg, d) matrix is view v of pair p: v ? z_j[p] : z_i[p]; positive = i^1.
 * a = z z^T / tau with the diagonal masked; loss = (1/M) sum_i [logsumexp_j a_ij - a_i,i^1], M = 2*Bg.
 * The similarity matrix lives in MFMA accumulators only.  A rank owning pairs [p0, p0+np) gets
 * loss_out[0] = sum over its 2*np rows / M, and dz_i/dz_j (np x d) = d(global mean loss)/dz for its pairs.
 * ws: float workspace of nsid_ntxent_ws_floats(Bg) elements. */
size_t nsid_ntxent_ws_floats(int Bg);
int nsid_ntxent_fwd_bwd(const float* z_i, const float* z_j, int Bg, int d, float tau, int p0, int np, float* ws,
                        float* loss_out, float* dz_i, float* dz_j, void* stream);

/* ---- optimiser (train.py:73-75: clip_grad_norm_(1.0) + Adam) -------------------------------------------
 * sumsq: partial[blocks] sums of squares of g (blocks = nsid_sumsq_blocks(n)).
 * adam:  norm = sqrt(sum partial); coef = min(1, max_norm/(norm+1e-6)); torch.optim.Adam update with g*coef.
 * hyper (device): [0]=lr, [1]=beta1, [2]=beta2, [3]=eps, [4]=max_norm (<=0: no clipping);
 * step (device int64) is incremented by the kernel; gnorm_out[0] = pre-clip norm. */
int nsid_sumsq_blocks(long n);
int nsid_sumsq_partial(const float* g, long n, float* partial, void* stream);
int nsid_adam_step(float* p, const float* g, float* m, float* v, long n, const float* hyper, int64_t* step,
                   const float* partial, int nblocks, float* gnorm_out, void* stream);

/* ---- step plumbing (train.py:54 zero_grad, loss hand-over, autograd's grad_output scaling): the captured step runs no
 * ATen kernel. fill_zero: p 16-byte aligned; scale: out[i] = x[i] * (scale ? scale[0] : 1), out may alias x. */
int nsid_fill_zero(void* p, size_t bytes, void* stream);
int nsid_scale_f32(const float* x, const float* scale, long n, float* out, void* stream);

/* ---- layout plumbing at the module boundary: (B, C, N) <-> node-major rows ------------------------------*/
int nsid_bcn_to_rows(const float* x, int B, int C, int N, void* rows, int ld, int rows_dtype, void* stream);
int nsid_rows_to_bcn(const void* rows, int ld, int B, int C, int N, float* x, int rows_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSID_H */

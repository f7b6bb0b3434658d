// What does a hipGraph charge for work that hangs OFF a dependent chain?  
//
// The training step is two chains (the views) of ~500 dependent kernels each; every layer also has a weight-gradient kernel
// that depends on ONE chain kernel and on which nothing depends until the end. Captured from two streams, those kernels sit IN
// the chain (stream order). Captured with an auxiliary stream per view and one event per layer they ran 45 % SLOWER in the real
// step. This microbenchmark separates "how the graph was built" from "what the hardware can do":
//   A  two streams, side kernels inline in stream order                      (what the step does today)
//   B  four streams: side kernels on an auxiliary stream per chain, one event edge per layer  (what lost)
//   C  explicit graph (hipGraphAddKernelNode): same DAG as B, no event nodes
//   D  explicit graph, chain kernels only                                     (lower bound: the side work is free)
//   E / F  A and B launched directly on streams, no graph (one host thread: launch cost included)
// Kernels are calibrated spin loops: `chain_us` on `chain_wgs` workgroups, `side_us` on `side_wgs`.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_dag_bench tools/graph_dag_bench.hip && /tmp/graph_dag_bench [layers chain_us side_us chain_wgs side_wgs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(long ticks, int* sink) {          // ticks of the 100 MHz constant clock
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {}
  if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && ticks < 0) *sink = 1;
}

struct Cfg { int layers; long chain_ticks, side_ticks; int chain_wgs, side_wgs; };

static void launch(hipStream_t s, long ticks, int wgs) { hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s, ticks, (int*)nullptr); }

static float time_graph(hipGraphExec_t ex, hipStream_t s, int reps) {
  CK(hipGraphLaunch(ex, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

static hipGraphExec_t capture(const Cfg& c, bool aux) {
  hipStream_t m, s2, a1, a2;
  CK(hipStreamCreate(&m)); CK(hipStreamCreate(&s2)); CK(hipStreamCreate(&a1)); CK(hipStreamCreate(&a2));
  hipEvent_t fork, join2, ja1, ja2;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join2, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ja1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ja2, hipEventDisableTiming));
  std::vector<hipEvent_t> ev(2 * c.layers);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CK(hipStreamBeginCapture(m, hipStreamCaptureModeGlobal));
  CK(hipEventRecord(fork, m));
  CK(hipStreamWaitEvent(s2, fork, 0));
  if (aux) { CK(hipStreamWaitEvent(a1, fork, 0)); CK(hipStreamWaitEvent(a2, fork, 0)); }
  for (int l = 0; l < c.layers; ++l) {
    hipStream_t chain[2] = {m, s2}, side[2] = {aux ? a1 : m, aux ? a2 : s2};
    for (int v = 0; v < 2; ++v) {
      launch(chain[v], c.chain_ticks, c.chain_wgs);
      if (aux) {
        CK(hipEventRecord(ev[2 * l + v], chain[v]));
        CK(hipStreamWaitEvent(side[v], ev[2 * l + v], 0));
      }
      launch(side[v], c.side_ticks, c.side_wgs);
    }
  }
  CK(hipEventRecord(join2, s2)); CK(hipStreamWaitEvent(m, join2, 0));
  if (aux) {
    CK(hipEventRecord(ja1, a1)); CK(hipStreamWaitEvent(m, ja1, 0));
    CK(hipEventRecord(ja2, a2)); CK(hipStreamWaitEvent(m, ja2, 0));
  }
  launch(m, 100, 1);                                        // the optimiser
  hipGraph_t g;
  CK(hipStreamEndCapture(m, &g));
  hipGraphExec_t ex;
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  return ex;
}

// the same work launched directly (no graph): host launch cost included, one host thread
static float time_streams(const Cfg& c, bool aux, int reps) {
  hipStream_t m, s2, a1, a2;
  CK(hipStreamCreate(&m)); CK(hipStreamCreate(&s2)); CK(hipStreamCreate(&a1)); CK(hipStreamCreate(&a2));
  std::vector<hipEvent_t> ev(2 * c.layers);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t j2, ja1, ja2, t0, t1;
  CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ja1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ja2, hipEventDisableTiming));
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  float best = 1e30f;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, m));
    CK(hipStreamWaitEvent(s2, t0, 0));
    if (aux) { CK(hipStreamWaitEvent(a1, t0, 0)); CK(hipStreamWaitEvent(a2, t0, 0)); }
    for (int l = 0; l < c.layers; ++l) {
      hipStream_t chain[2] = {m, s2}, side[2] = {aux ? a1 : m, aux ? a2 : s2};
      for (int v = 0; v < 2; ++v) {
        launch(chain[v], c.chain_ticks, c.chain_wgs);
        if (aux) {
          CK(hipEventRecord(ev[2 * l + v], chain[v]));
          CK(hipStreamWaitEvent(side[v], ev[2 * l + v], 0));
        }
        launch(side[v], c.side_ticks, c.side_wgs);
      }
    }
    CK(hipEventRecord(j2, s2)); CK(hipStreamWaitEvent(m, j2, 0));
    if (aux) {
      CK(hipEventRecord(ja1, a1)); CK(hipStreamWaitEvent(m, ja1, 0));
      CK(hipEventRecord(ja2, a2)); CK(hipStreamWaitEvent(m, ja2, 0));
    }
    launch(m, 100, 1);
    CK(hipEventRecord(t1, m));
    CK(hipStreamSynchronize(m));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, t0, t1));
    if (r > 0 && ms < best) best = ms;
  }
  return best * 1e3f;
}

static hipGraphExec_t explicit_graph(const Cfg& c, bool with_side) {
  hipGraph_t g;
  CK(hipGraphCreate(&g, 0));
  static long chain_t, side_t, one_t = 100;
  static int* nullsink = nullptr;
  chain_t = c.chain_ticks; side_t = c.side_ticks;
  auto add = [&](long* ticks, int wgs, const std::vector<hipGraphNode_t>& deps) {
    void* args[2] = {ticks, &nullsink};
    hipKernelNodeParams p{};
    p.func = reinterpret_cast<void*>(spin_kernel);
    p.gridDim = dim3(wgs); p.blockDim = dim3(256); p.sharedMemBytes = 0; p.kernelParams = args; p.extra = nullptr;
    hipGraphNode_t n;
    CK(hipGraphAddKernelNode(&n, g, deps.data(), deps.size(), &p));
    return n;
  };
  std::vector<hipGraphNode_t> last(2), tails;
  bool have[2] = {false, false};
  for (int l = 0; l < c.layers; ++l)
    for (int v = 0; v < 2; ++v) {
      std::vector<hipGraphNode_t> d;
      if (have[v]) d.push_back(last[v]);
      last[v] = add(&chain_t, c.chain_wgs, d);
      have[v] = true;
      if (with_side) tails.push_back(add(&side_t, c.side_wgs, {last[v]}));
    }
  tails.push_back(last[0]); tails.push_back(last[1]);
  add(&one_t, 1, tails);
  hipGraphExec_t ex;
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  return ex;
}

int main(int argc, char** argv) {
  Cfg c{100, 1000, 1500, 256, 256};                         // ticks = 10 ns: 10 us chain kernels, 15 us side kernels
  if (argc > 1) c.layers = atoi(argv[1]);
  if (argc > 2) c.chain_ticks = atol(argv[2]) * 100;
  if (argc > 3) c.side_ticks = atol(argv[3]) * 100;
  if (argc > 4) c.chain_wgs = atoi(argv[4]);
  if (argc > 5) c.side_wgs = atoi(argv[5]);
  hipStream_t s;
  CK(hipStreamCreate(&s));
  printf("layers %d per chain (2 chains), chain kernel %.0f us x %d workgroups, side kernel %.0f us x %d workgroups\n", c.layers,
         c.chain_ticks / 100.0, c.chain_wgs, c.side_ticks / 100.0, c.side_wgs);
  const double serial = c.layers * (c.chain_ticks + c.side_ticks) / 100.0, chain_only = c.layers * c.chain_ticks / 100.0;
  printf("  kernel time per chain: inline %.0f us, chain alone %.0f us\n", serial, chain_only);
  printf("  A two streams, side kernels inline        : %8.1f us per replay\n", time_graph(capture(c, false), s, 10));
  printf("  B four streams, event edge per layer      : %8.1f us per replay\n", time_graph(capture(c, true), s, 10));
  printf("  C explicit DAG, side kernels hang off     : %8.1f us per replay\n", time_graph(explicit_graph(c, true), s, 10));
  printf("  D explicit DAG, chains only               : %8.1f us per replay\n", time_graph(explicit_graph(c, false), s, 10));
  printf("  E no graph, two streams inline            : %8.1f us (best of 5, host launches included)\n", time_streams(c, false, 5));
  printf("  F no graph, four streams + events         : %8.1f us (best of 5, host launches included)\n", time_streams(c, true, 5));
  return 0;
}

// Step engine: replays a captured step as plain stream launches with its own dependency plan.
//
// Why: on this ROCm a hipGraph pays 15-20 us for every node that forks off a dependent chain (tools/graph_dag_bench.hip), so the
// captured step keeps every layer's weight gradient INSIDE the view's chain although nothing waits for it before the optimiser.
// Launched directly on streams the same fork costs an event record + wait (~2 us, same microbenchmark, variants E / F). The engine
// takes the hipGraph torch captured (kernel nodes, their arguments and the true + stream-order edges), drops the edges that lead
// OUT of weight-gradient kernels (their only consumer is the optimiser tail; the caller guarantees that no buffer they read is
// freed or overwritten before the step ends: functional.KEEP), and replays the nodes in capture order on four streams: the two
// view chains, and one auxiliary stream per chain for the floating kernels; cross-stream edges become events.
//
// Host cost per replay: one hipLaunchKernel per node plus ~2 event calls per floating kernel (about 1 600 calls, ~4 ms, on the
// calling thread; the GPU step is ~8 ms, so the host stays ahead).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "nsid_common.h"

namespace {

struct Node {
  hipGraphNodeType type;
  hipKernelNodeParams kp;
  hipMemsetParams ms;
  std::string name;
  std::vector<int> deps, succs;
  bool floating = false;       // weight gradient (or its unpack pass): only the optimiser tail waits for it
  int stream = 0;
  int event = -1;              // index of the event recorded after this node (when another stream waits for it)
  std::vector<int> waits;      // events this node's stream waits for before the launch
};

struct Engine {
  std::vector<Node> nodes;
  std::vector<hipStream_t> streams;      // [0] = the caller's stream (set per replay), [1..3] owned
  std::vector<hipEvent_t> events;
  hipEvent_t start = nullptr;
  std::vector<hipEvent_t> tail;          // one per owned stream: joined into the caller's stream at the end
  int n_floating = 0, n_events = 0, n_kernels = 0;
};

bool is_wgrad(const std::string& n) {
  if (n.find("wgrad3_kernel") != std::string::npos) return true;
  const size_t p = n.find("gemm_kernelILi");
  if (p == std::string::npos) return false;
  // gemm_kernel<BM, BN, A_RMAJOR, B_RMAJOR, ...>: ILi<BM>ELi<BN>ELb<A>ELb<B>E...; the weight gradient is the only <false, false>
  const size_t q = n.find("ELb", p);
  return q != std::string::npos && n.compare(q, 9, "ELb0ELb0E") == 0;
}
bool rides_with_wgrad(const std::string& n) { return n.find("unpack_ds_wgrad") != std::string::npos; }
bool is_tail_start(const std::string& n) { return n.find("sumsq_kernel") != std::string::npos; }

}  // namespace

extern "C" int nsid_engine_build(void* graph_, void** out, int float_wgrad, char* log, size_t log_len) {
  auto say = [&](const char* m) { if (log && log_len) snprintf(log, log_len, "%s", m); };
  if (!graph_ || !out) return NSID_EINVAL;
  hipGraph_t graph = static_cast<hipGraph_t>(graph_);
  size_t nn = 0, ne = 0;
  if (hipGraphGetNodes(graph, nullptr, &nn) != hipSuccess || nn == 0) { say("hipGraphGetNodes failed"); return NSID_EINVAL; }
  std::vector<hipGraphNode_t> gn(nn);
  if (hipGraphGetNodes(graph, gn.data(), &nn) != hipSuccess) { say("hipGraphGetNodes failed"); return NSID_EINVAL; }
  if (hipGraphGetEdges(graph, nullptr, nullptr, &ne) != hipSuccess) { say("hipGraphGetEdges failed"); return NSID_EINVAL; }
  std::vector<hipGraphNode_t> ef(ne), et(ne);
  if (ne && hipGraphGetEdges(graph, ef.data(), et.data(), &ne) != hipSuccess) { say("hipGraphGetEdges failed"); return NSID_EINVAL; }
  auto* e = new Engine();
  e->nodes.resize(nn);
  auto index_of = [&](hipGraphNode_t n) { for (size_t i = 0; i < nn; ++i) if (gn[i] == n) return (int)i; return -1; };
  // (nn ~ 1 000: the quadratic lookup below runs once per engine)
  std::vector<std::pair<int, int>> edges;
  for (size_t i = 0; i < ne; ++i) edges.emplace_back(index_of(ef[i]), index_of(et[i]));
  for (size_t i = 0; i < nn; ++i) {
    Node& n = e->nodes[i];
    if (hipGraphNodeGetType(gn[i], &n.type) != hipSuccess) { say("hipGraphNodeGetType failed"); delete e; return NSID_EINVAL; }
    if (n.type == hipGraphNodeTypeKernel) {
      if (hipGraphKernelNodeGetParams(gn[i], &n.kp) != hipSuccess) { say("hipGraphKernelNodeGetParams failed"); delete e; return NSID_EINVAL; }
      const char* nm = hipKernelNameRefByPtr(n.kp.func, nullptr);
      n.name = nm ? nm : "";
      ++e->n_kernels;
    } else if (n.type == hipGraphNodeTypeMemset) {
      if (hipGraphMemsetNodeGetParams(gn[i], &n.ms) != hipSuccess || n.ms.height > 1 ||
          (n.ms.elementSize != 1 && n.ms.elementSize != 4)) {
        say("memset node of an unsupported shape");
        delete e;
        return NSID_EINVAL;
      }
      n.name = "memset";
    } else if (n.type != hipGraphNodeTypeEmpty) {
      char buf[96];
      snprintf(buf, sizeof buf, "node %zu has type %d: only kernel, 1-D memset and empty nodes are replayed", i, (int)n.type);
      say(buf);
      delete e;
      return NSID_EINVAL;
    }
  }
  // replay order = a topological order that follows the creation (capture) order wherever the edges allow it
  {
    for (auto& ed : edges)
      if (ed.first < 0 || ed.second < 0) { say("edge to a node outside the graph"); delete e; return NSID_EINVAL; }
    bool forward = true;
    for (auto& ed : edges) forward = forward && ed.first < ed.second;
    if (!forward) {
      std::vector<int> indeg(nn, 0), order, pos(nn, 0);
      for (auto& ed : edges) ++indeg[ed.second];
      std::vector<char> done(nn, 0);
      for (size_t placed = 0; placed < nn; ++placed) {
        int pick = -1;
        for (size_t i = 0; i < nn; ++i) if (!done[i] && indeg[i] == 0) { pick = (int)i; break; }
        if (pick < 0) { say("the graph has a cycle"); delete e; return NSID_EINVAL; }
        done[pick] = 1; pos[pick] = (int)order.size(); order.push_back(pick);
        for (auto& ed : edges) if (ed.first == pick) --indeg[ed.second];
      }
      std::vector<Node> sorted(nn);
      for (size_t i = 0; i < nn; ++i) sorted[i] = e->nodes[order[i]];
      e->nodes.swap(sorted);
      for (auto& ed : edges) { ed.first = pos[ed.first]; ed.second = pos[ed.second]; }
    }
  }
  int tail0 = -1;
  for (size_t i = 0; i < nn; ++i)
    if (is_tail_start(e->nodes[i].name)) { tail0 = (int)i; break; }
  if (float_wgrad && tail0 >= 0) {
    for (size_t i = 0; i < nn; ++i)
      if ((int)i < tail0 && is_wgrad(e->nodes[i].name)) e->nodes[i].floating = true;
    // a pass that consumes the gradient at once (the Downsample unpack) floats with its weight gradient
    for (bool changed = true; changed;) {
      changed = false;
      for (auto& ed : edges)
        if (e->nodes[ed.first].floating && !e->nodes[ed.second].floating && rides_with_wgrad(e->nodes[ed.second].name)) {
          e->nodes[ed.second].floating = true;
          changed = true;
        }
    }
    // edges out of a floating node: kept between floating nodes, otherwise replaced by one edge to the optimiser tail
    std::vector<std::pair<int, int>> kept;
    for (auto& ed : edges)
      if (!e->nodes[ed.first].floating || e->nodes[ed.second].floating) kept.push_back(ed);
    for (size_t i = 0; i < nn; ++i)
      if (e->nodes[i].floating) { kept.emplace_back((int)i, tail0); ++e->n_floating; }
    // a node that followed a floating one in its stream still needs what that floating node waited for: it inherits the nearest
    // NON-floating ancestors through any run of floating nodes (a weight gradient and its unpack pass are two in a row)
    std::vector<std::pair<int, int>> extra;
    for (auto& ed : edges) {
      if (!e->nodes[ed.first].floating || e->nodes[ed.second].floating || ed.second == tail0) continue;
      std::vector<int> walk{ed.first};
      while (!walk.empty()) {
        const int f = walk.back();
        walk.pop_back();
        for (auto& in : edges) {
          if (in.second != f) continue;
          if (e->nodes[in.first].floating) walk.push_back(in.first);
          else extra.emplace_back(in.first, ed.second);
        }
      }
    }
    for (auto& x : extra) kept.push_back(x);
    edges.swap(kept);
  }
  for (auto& ed : edges) { e->nodes[ed.second].deps.push_back(ed.first); e->nodes[ed.first].succs.push_back(ed.second); }
  // ---- streams: 0 / 1 = the two chains (recovered from the capture order), 2 / 3 = their auxiliaries
  int tails[2] = {-1, -1};
  long last_use[2] = {-1, -1};
  for (size_t i = 0; i < nn; ++i) {
    Node& n = e->nodes[i];
    if (n.floating) {
      int c = 0;
      for (int d : n.deps) { c = e->nodes[d].stream; break; }
      n.stream = c < 2 ? 2 + c : c;                 // (the unpack pass follows its wgrad: same auxiliary)
      continue;
    }
    int s = -1;
    for (int d : n.deps)
      for (int k = 0; k < 2; ++k)
        if (d == tails[k] && (s < 0 || k < s)) s = k;
    if (s < 0) s = n.deps.empty() ? 0 : (last_use[0] <= last_use[1] ? 0 : 1);   // a fork: the chain that has been idle longest
    n.stream = s;
    tails[s] = (int)i;
    last_use[s] = (long)i;
  }
  // ---- events for cross-stream edges
  for (size_t i = 0; i < nn; ++i) {
    Node& n = e->nodes[i];
    for (int d : n.deps) {
      Node& p = e->nodes[d];
      if (p.stream == n.stream) continue;           // stream order covers it (capture order = launch order)
      if (p.event < 0) p.event = e->n_events++;
      n.waits.push_back(p.event);
    }
  }
  bool ok = true;
  e->events.resize(e->n_events);
  for (auto& ev : e->events) ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && ok;
  e->streams.assign(4, nullptr);
  for (int k = 1; k < 4; ++k) ok = hipStreamCreateWithFlags(&e->streams[k], hipStreamNonBlocking) == hipSuccess && ok;
  ok = hipEventCreateWithFlags(&e->start, hipEventDisableTiming) == hipSuccess && ok;
  e->tail.resize(3);
  for (auto& ev : e->tail) ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && ok;
  if (!ok) { say("stream / event creation failed"); delete e; return NSID_ELAUNCH; }
  if (log && log_len)
    snprintf(log, log_len, "%d kernel nodes of %zu, %zu edges, %d floating, %d events, optimiser tail at node %d", e->n_kernels, nn,
             edges.size(), e->n_floating, e->n_events, tail0);
  *out = e;
  return NSID_OK;
}

extern "C" int nsid_engine_replay(void* handle, void* main_stream) {
  if (!handle) return NSID_EINVAL;
  Engine* e = static_cast<Engine*>(handle);
  e->streams[0] = static_cast<hipStream_t>(main_stream);
  (void)hipGetLastError();
  hipError_t err = hipSuccess;
  auto chk = [&](hipError_t r) { if (r != hipSuccess && err == hipSuccess) err = r; };
  chk(hipEventRecord(e->start, e->streams[0]));
  for (int k = 1; k < 4; ++k) chk(hipStreamWaitEvent(e->streams[k], e->start, 0));
  for (Node& n : e->nodes) {
    hipStream_t s = e->streams[n.stream];
    for (int w : n.waits) chk(hipStreamWaitEvent(s, e->events[w], 0));
    if (n.type == hipGraphNodeTypeKernel)
      chk(hipLaunchKernel(n.kp.func, n.kp.gridDim, n.kp.blockDim, n.kp.kernelParams, n.kp.sharedMemBytes, s));
    else if (n.type == hipGraphNodeTypeMemset)
      chk(n.ms.elementSize == 4 ? hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(n.ms.dst), (int)n.ms.value, n.ms.width, s)
                                : hipMemsetAsync(n.ms.dst, (int)n.ms.value, n.ms.width, s));
    if (n.event >= 0) chk(hipEventRecord(e->events[n.event], s));
  }
  for (int k = 1; k < 4; ++k) {
    chk(hipEventRecord(e->tail[k - 1], e->streams[k]));
    chk(hipStreamWaitEvent(e->streams[0], e->tail[k - 1], 0));
  }
  if (err != hipSuccess) {
    fprintf(stderr, "[nsid] engine replay failed: %s\n", hipGetErrorString(err));
    return NSID_ELAUNCH;
  }
  return nsid_launch_status();
}

extern "C" int nsid_engine_destroy(void* handle) {
  if (!handle) return NSID_OK;
  Engine* e = static_cast<Engine*>(handle);
  for (int k = 1; k < 4; ++k) if (e->streams[k]) { (void)hipStreamSynchronize(e->streams[k]); (void)hipStreamDestroy(e->streams[k]); }
  for (auto& ev : e->events) (void)hipEventDestroy(ev);
  for (auto& ev : e->tail) (void)hipEventDestroy(ev);
  if (e->start) (void)hipEventDestroy(e->start);
  delete e;
  return NSID_OK;
}

"""Step engine (csrc/engine.hip): a captured step replayed as plain stream launches with its own dependency plan.

torch captures the step (that fixes every buffer address and yields the kernel nodes with their arguments); the engine then
replays those nodes on four streams — the two view chains plus an auxiliary stream per chain onto which the weight-gradient
kernels float, waited for only by the optimiser tail. A hipGraph cannot do that cheaply on this ROCm: every node that forks off a
chain costs 15-20 us there, an event on a stream ~2 us (tools/graph_dag_bench.hip).

Safety contract: a floating weight gradient reads dr and the saved activation of its layer long after the chain has moved on, so
NOTHING allocated during the captured step may be freed (and its memory reused inside the step) before the step ends.
`retain_allocations()` keeps every tensor that ops.py allocates during the capture alive for the lifetime of the engine."""
import contextlib
import ctypes

import torch

from ._lib import lib


@contextlib.contextmanager
def retain_allocations(keep: list):
    """every torch.empty / torch.empty_like result created inside is appended to `keep` (the capture's private pool then never hands
    the same block to two tensors of the step)"""
    real_empty, real_like = torch.empty, torch.empty_like

    def empty(*a, **k):
        t = real_empty(*a, **k)
        keep.append(t)
        return t

    def empty_like(*a, **k):
        t = real_like(*a, **k)
        keep.append(t)
        return t

    torch.empty, torch.empty_like = empty, empty_like
    try:
        yield keep
    finally:
        torch.empty, torch.empty_like = real_empty, real_like


class StepEngine:
    def __init__(self, graph: "torch.cuda.CUDAGraph", keep: list, float_wgrad: bool = True):
        """graph: captured with torch.cuda.CUDAGraph(keep_graph=True) under retain_allocations(keep)"""
        raw = graph.raw_cuda_graph()
        out = ctypes.c_void_p()
        log = ctypes.create_string_buffer(512)
        rc = lib.nsid_engine_build(ctypes.c_void_p(int(raw)), ctypes.byref(out), int(float_wgrad), log, len(log))
        self.info = log.value.decode()
        if rc != 0:
            raise RuntimeError(f"nsid_engine_build refused the graph: {self.info}")
        self._h, self.graph, self.keep = out, graph, keep

    def replay(self) -> None:
        rc = lib.nsid_engine_replay(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError("nsid_engine_replay failed")

    def close(self) -> None:
        if self._h is not None:
            lib.nsid_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""hipGraph capture of fork/join patterns: a fork from the capture stream works; a NESTED fork (from a stream that is itself
an unjoined fork) crashes hipStreamEndCapture on ROCm 7.0 / torch 2.10 (found while trying deferred weight-gradient GEMMs on parallel branches).
Usage: python tools/fork_capture_test.py"""
import sys, os, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsampleid_amd import ops
ops.set_gemm_precision("bf16")
dev = "cuda"
M, N, K = 16384, 256, 256
x = torch.randn(M, K, device=dev).bfloat16(); d = torch.randn(M, N, device=dev).bfloat16()
dw = torch.zeros(N, K, device=dev)
pool = [torch.cuda.Stream() for _ in range(2)]
view = torch.cuda.Stream()
def body(nested):
    cur = torch.cuda.current_stream()
    def inner(st):
        for p in pool:
            p.wait_stream(st)
            with torch.cuda.stream(p):
                for _ in range(3):
                    ops.linear_bwd_weight(d, x, dw, M, N, K)
        for p in pool:
            st.wait_stream(p)
    if nested:
        view.wait_stream(cur)
        with torch.cuda.stream(view):
            ops.linear_bwd_weight(d, x, dw, M, N, K)
            inner(view)
        cur.wait_stream(view)
    else:
        inner(cur)
for nested in (False, True):
    body(nested); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(nested)
    g.replay(); torch.cuda.synchronize()
    print("ok nested", nested, float(dw.abs().sum()))

"""hipGraph capture of fork/join patterns: a fork from the capture stream works; a NESTED fork (from a stream that is itself
an unjoined fork) crashes hipStreamEndCapture on ROCm 7.0 / torch 2.10 (found while trying deferred weight-gradient GEMMs on parallel branches).
Usage: python tools/fork_capture_test.py            the safe fork pattern only (prints PASS/FAIL, exit code 0/1)
       python tools/fork_capture_test.py --nested   ALSO the nested fork, in a CHILD process whose exit code is reported:
                                                    it is known to crash the capturing process, never run it in a process
                                                    whose GPU state matters (and not at all on a shared pool unless asked)"""
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
def run(nested):
    body(nested); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(nested)
    before = float(dw.abs().sum())
    g.replay(); torch.cuda.synchronize()
    return float(dw.abs().sum()) > before


if "--child-nested" in sys.argv:                 # the crashing pattern, isolated in its own process
    sys.exit(0 if run(True) else 1)
ok = run(False)
print("fork from the capture stream:", "PASS" if ok else "FAIL")
if "--nested" in sys.argv:
    import subprocess
    rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-nested"]).returncode
    print(f"nested fork (child process): exit code {rc}" + (" (crashed, as on ROCm 7.0 / torch 2.10)" if rc != 0 else " (works on this runtime)"))
sys.exit(0 if ok else 1)

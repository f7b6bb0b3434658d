import os, sys, runpy
sys.path.insert(0, os.getcwd())
import torch
from neuralsampleid_amd import parallel
orig = parallel.GradReducer._fire
def _fire(self, b):
    import threading
    print("fire", b, "thread", threading.current_thread().name, "cur", torch.cuda.current_stream(), "capturing", torch.cuda.is_current_stream_capturing(), file=sys.stderr, flush=True)
    return orig(self, b)
parallel.GradReducer._fire = _fire
sys.argv = ["bench.py", "--steps", "5", "--warmup", "3", "--no-cpu-baseline", "--no-roofline"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")

import sys, os, torch
sys.path.insert(0, os.getcwd())
from neuralsampleid_amd import ops
for Bg, npairs in ((256, 256), (2048, 256), (2048, 2048)):
    g = torch.Generator().manual_seed(0)
    zi = torch.nn.functional.normalize(torch.randn(Bg, 128, generator=g)).cuda()
    zj = torch.nn.functional.normalize(zi.cpu() + 0.3 * torch.randn(Bg, 128, generator=g)).cuda()
    fn = lambda: ops.ntxent_fwd_bwd(zi, zj, 0.05, 0, npairs)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(10): fn()
    gph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
    print(f"Bg={Bg} own pairs={npairs}: {1e3*e0.elapsed_time(e1)/10:.1f} us per fwd+bwd")

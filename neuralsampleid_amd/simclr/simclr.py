"""Mirror of simclr/simclr.py::SimCLR (reference :7-47): forward(x_i, x_j) -> (h_i, h_j, z_i, z_j)."""
import torch
import torch.nn as nn

from .. import functional as F_
from .. import ops
from ..encoder.gcn_lib.torch_vertex import _split
from ..encoder.graph_encoder import GraphEncoder
from ..peak_extractor import GPUPeakExtractorv2


class SimCLR(nn.Module):
    def __init__(self, cfg, encoder, overlap_views=False):
        """overlap_views (extension, default off): run the two views on two HIP streams (functional.ViewOrder)."""
        super().__init__()
        self.encoder = encoder
        self.cfg = cfg
        self.overlap_views = overlap_views
        self._side_stream = None
        d, h, u = cfg["d"], cfg["h"], cfg["u"]
        if cfg["arch"] != "grafp":
            raise NotImplementedError("only arch='grafp' is on the MI355X path (resnet-ibn is a different model family)")
        self.peak_extractor = GPUPeakExtractorv2(cfg)
        self.projector = nn.Sequential(nn.Linear(h, d * u), nn.ELU(), nn.Linear(d * u, d))

    def _project(self, h):
        params, buffers = _split(self.projector)
        return F_.run_block(F_.projector_forward, F_.projector_backward, params, buffers, h)

    def _embed(self, x):
        if isinstance(self.encoder, GraphEncoder):      # node-major all the way: no layout round trip
            B, H, W = x.shape
            N = (H // self.peak_extractor.patch_bins) * (W // self.peak_extractor.patch_frames)
            h = self.encoder.forward_rows(self.peak_extractor.forward_rows(x), B, N)
        else:
            h = self.encoder(self.peak_extractor(x))
        return h, self._project(h)

    def forward(self, x_i, x_j):
        if self.overlap_views and x_i.is_cuda and self.training:
            return self._forward_two_streams(x_i, x_j)
        h_i, z_i = self._embed(x_i)     # the encoder runs once per view: BatchNorm statistics are per view
        h_j, z_j = self._embed(x_j)
        return h_i, h_j, z_i, z_j

    def _forward_two_streams(self, x_i, x_j):
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=x_i.device)
            F_.SIDE_STREAMS.append(self._side_stream)
        side = self._side_stream
        side.wait_stream(main)                       # inputs/weights written on the main stream are visible
        vo = F_.VIEW_ORDER
        vo.mode, vo.pending = "a", {"a": [], "b": []}
        try:
            h_i, z_i = self._embed(x_i)
            vo.mode = "b"
            with torch.cuda.stream(side):
                h_j, z_j = self._embed(x_j)
        finally:
            vo.mode = None
        main.wait_stream(side)
        # running statistics: view i's update, then view j's, for every BatchNorm layer at once (see functional.ViewOrder)
        ops.bn_running_update(vo.pending["a"], vo.pending["b"])
        vo.pending = {"a": [], "b": []}
        for t in (h_j, z_j):                         # produced on the side stream, consumed on the main one
            t.record_stream(main)
        return h_i, h_j, z_i, z_j

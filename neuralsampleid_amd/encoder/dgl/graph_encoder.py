"""Signature shim for encoder/dgl/graph_encoder.py::GraphEncoderDGL (reference :67-147) — SURVEY.md §8f-1.

train.py:111, test_fp.py:235/266/301, downstream.py:112/159 and ablation.py:88/161 build `GraphEncoderDGL(cfg=cfg,
in_channels=..., k=..., size=...)` and call `encoder(x)` or `encoder(x, return_pre_proj=True) -> (x_nodes, x_emb)`.
This class gives those call sites the MI355X kernels with NO DGL runtime (north_star).  It is the gcn_lib encoder
(GraphEncoder: dense kNN on L2-normalised features, max-relative aggregation) under the DGL constructor/forward
signature; two documented differences from the reference's DGL variant:
  * the graph blocks are USED — the reference's `_apply_graph_block` computes them and returns its input unchanged
    (reference :149-160), so its output never depends on the graph;
  * the module tree / state_dict keys are GraphEncoder's, so DGL-variant checkpoints (model_tc_35_best.pth) do not load.
`include_self` is accepted for signature compatibility: the dense kNN always contains the self edge (rank 0)."""
import torch

from ... import functional as F_
from ..graph_encoder import GraphEncoder


class GraphEncoderDGL(GraphEncoder):
    def __init__(self, cfg=None, k=3, conv="mr", act="relu", norm="batch", bias=True, dropout=0.0, dilation=True,
                 epsilon=0.2, drop_path=0.1, size="t", emb_dims=1024, in_channels=3, include_self=False, **kw):
        if cfg is None:
            raise ValueError("cfg is required (n_mels, n_frames, patch_bins, patch_frames)")
        super().__init__(cfg, k=k, conv=conv, act=act, norm=norm, bias=bias, dropout=dropout, dilation=dilation,
                         epsilon=epsilon, drop_path=drop_path, size=size, emb_dims=emb_dims, in_channels=in_channels,
                         **kw)
        self.include_self = include_self

    def forward(self, x, return_pre_proj=False):
        """x (B, C, N) -> x_emb (B, emb_dims), or (x_nodes (B, C_last, N_last), x_emb) with return_pre_proj=True"""
        B, C, N = x.shape
        out = self.forward_rows(F_.to_rows(x), B, N, return_nodes=return_pre_proj)
        if not return_pre_proj:
            return out
        rows, n_last, emb = out
        return F_.from_rows(rows, B, n_last), emb

"""Mirror of encoder/gcn_lib/torch_vertex.py (reference :11-34, :92-139, :142-195): MRConv2d, DyGraphConv2d, Grapher.

`forward` keeps the reference's (B, C, N, 1) signature; `forward_rows` is the node-major entry GraphEncoder uses so
that a whole encoder pass changes layout once on the way in and never again."""
import torch
from torch import nn

from ... import functional as F_
from ... import ops
from .pos_embed import relative_pos_table
from .torch_edge import DenseDilatedKnnGraph
from .torch_nn import BasicConv


def _split(module):
    params = {n: p for n, p in module.named_parameters() if p.requires_grad}
    buffers = dict(module.named_buffers())
    return params, buffers


def mrconv_forward(x, P, S, B, N, idx, training):
    M, C = x.shape
    u, amax = ops.mr_aggregate_fwd(x, idx, B, N, C, None, want_argmax=S is not None)
    r, aff = F_.conv_bn(u, M, C // 2, P["nn.0.weight"].shape[0] // 4, P["nn.0.weight"], P.get("nn.0.bias"),
                        F_._bn(P, S, "nn.1."), training, groups=4)
    out = ops.bn_apply(r, aff, ops.ACT_RELU)
    if S is not None:
        S.update(u=u, amax=amax, r=r, aff=aff, idx=idx, B=B, N=N, C=C)
    return out


def mrconv_backward(dout, P, S, G):
    u, amax, r, aff, idx, B, N, C = (S[k] for k in ("u", "amax", "r", "aff", "idx", "B", "N", "C"))
    M, Co = r.shape
    dr = ops.bn_backward(dout, r, aff, ops.ACT_RELU, G["nn.1.weight"], G["nn.1.bias"])
    if "nn.0.bias" in G:
        F_._bias_grad_before_bn(dr, G["nn.0.bias"])
    ops.linear_bwd_weight(dr, u, ops.w2d(G["nn.0.weight"]), M, Co // 4, C // 2, 4)
    du = ops.linear_bwd_data(dr, ops.w2d(P["nn.0.weight"]), M, Co // 4, C // 2, 4)
    return ops.mr_aggregate_bwd(du, idx, amax, B, N, C)


class MRConv2d(nn.Module):
    """Max-relative graph conv: forward(x (B,C,N,1), edge_index (2,B,N,k)) -> (B,2C,N,1) (reference :11-34)."""

    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super().__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index, y=None):
        if y is not None:
            raise NotImplementedError("y (r > 1) is unreachable from GraphEncoder")
        B, C, N = x.shape[0], x.shape[1], x.shape[2]
        idx = edge_index[0].to(torch.int32).contiguous()
        params, buffers = _split(self)
        out = F_.run_block(mrconv_forward, mrconv_backward, params, buffers, F_.to_rows(x), B, N, idx, self.training)
        return F_.from_rows(out, B, N).unsqueeze(-1)


class GraphConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, conv="edge", act="relu", norm=None, bias=True):
        super().__init__()
        if conv != "mr":
            # 'edge' / 'sage' / 'gin' exist in the reference but GraphEncoder hard-codes 'mr' (graph_encoder.py:143)
            raise NotImplementedError("conv:{} is not supported".format(conv))
        self.gconv = MRConv2d(in_channels, out_channels, act, norm, bias)

    def forward(self, x, edge_index, y=None):
        return self.gconv(x, edge_index, y)


class DyGraphConv2d(GraphConv2d):
    """kNN graph rebuilt on every forward, then MRConv2d (reference :114-139)."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None,
                 bias=True, stochastic=False, epsilon=0.0, r=1):
        super().__init__(in_channels, out_channels, conv, act, norm, bias)
        if r != 1:
            raise NotImplementedError("r > 1 is unreachable from GraphEncoder (graph_encoder.py:171)")
        self.k, self.d, self.r = kernel_size, dilation, r
        self.dilated_knn_graph = DenseDilatedKnnGraph(kernel_size, dilation, stochastic, epsilon)

    def forward(self, x, relative_pos=None):
        B, C, H, W = x.shape
        x = x.reshape(B, C, -1, 1).contiguous()
        edge_index = self.dilated_knn_graph(x, None, relative_pos)
        x = super().forward(x, edge_index, None)
        return x.reshape(B, -1, H, W).contiguous()


class Grapher(nn.Module):
    """fc1 -> dynamic max-relative graph conv -> fc2 -> + residual (reference :142-195)."""

    def __init__(self, in_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None, bias=True,
                 stochastic=False, epsilon=0.0, r=1, n=196, drop_path=0.0, relative_pos=False):
        super().__init__()
        if drop_path > 0.0:
            raise NotImplementedError("DropPath is Identity in the reference as shipped (graph_encoder.py:161-173)")
        self.channels, self.n, self.r = in_channels, n, r
        self.fc1 = nn.Sequential(nn.Conv2d(in_channels, in_channels, 1, stride=1, padding=0),
                                 nn.BatchNorm2d(in_channels))
        self.graph_conv = DyGraphConv2d(in_channels, in_channels * 2, kernel_size, dilation, conv, act, norm, bias,
                                        stochastic, epsilon, r)
        self.fc2 = nn.Sequential(nn.Conv2d(in_channels * 2, in_channels, 1, stride=1, padding=0),
                                 nn.BatchNorm2d(in_channels))
        self.drop_path = nn.Identity()
        self.relative_pos = None
        if relative_pos:
            # state_dict compatibility only: the reference builds a sin-cos table here (torch_vertex.py:165-172) and
            # never reads it (forward passes relative_pos=None, :189). Same key, shape, values, requires_grad=False.
            self.relative_pos = nn.Parameter(relative_pos_table(in_channels, n, r), requires_grad=False)

    def forward_rows(self, rows, B, N):
        params, buffers = _split(self)
        return F_.run_block(F_.grapher_forward, F_.grapher_backward, params, buffers, rows, B, N,
                            self.graph_conv.k, self.graph_conv.d, self.training)

    def forward(self, x):
        B, C, N = x.shape[0], x.shape[1], x.shape[2]
        return F_.from_rows(self.forward_rows(F_.to_rows(x), B, N), B, N).unsqueeze(-1)

"""Mirror of encoder/graph_encoder.py (reference :38-50 Downsample, :67-89 FFN, :91-214 GraphEncoder).

Same constructor signatures, module tree and therefore `state_dict` keys as the reference; the forward pass runs on
the gfx950 kernels in node-major layout (functional.py)."""
import torch
import torch.nn as nn
from torch.nn import Sequential as Seq

from .. import functional as F_
from .gcn_lib.torch_nn import act_layer
from .gcn_lib.torch_vertex import Grapher, _split

SIZES = {"t": ([2, 2, 6, 2], [64, 128, 256, 512]), "s": ([2, 2, 6, 2], [80, 160, 400, 640]),
         "m": ([2, 2, 16, 2], [96, 192, 384, 768])}
SIZE_DEFAULT = ([2, 2, 18, 2], [128, 256, 512, 1024])


class Downsample(nn.Module):
    """Conv2d 3x3 stride 2 pad 1 + BatchNorm2d on a (B,C,N,1) map: N -> N/2, no activation (reference :38-50)."""

    def __init__(self, in_dim=3, out_dim=768):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_dim, out_dim, 3, stride=2, padding=1), nn.BatchNorm2d(out_dim))

    def forward_rows(self, rows, B, N):
        params, buffers = _split(self)
        return F_.run_block(F_.downsample_forward, F_.downsample_backward, params, buffers, rows, B, N, self.training)

    def forward(self, x):
        B, C, N = x.shape[0], x.shape[1], x.shape[2]
        if x.dim() != 4 or x.shape[3] != 1:
            raise NotImplementedError("Downsample on the GraFP path sees width-1 maps (B,C,N,1)")
        out = self.forward_rows(F_.to_rows(x), B, N)
        return F_.from_rows(out, B, (N - 1) // 2 + 1).unsqueeze(-1)


class FFN(nn.Module):
    """conv1x1+BN -> ReLU -> conv1x1+BN -> + shortcut (reference :67-89)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act="relu", drop_path=0.0):
        super().__init__()
        out_features = out_features if out_features is not None else in_features
        hidden_features = hidden_features if hidden_features is not None else in_features
        if drop_path > 0.0 or act != "relu" or out_features != in_features:
            raise NotImplementedError("the MI355X FFN implements act='relu', drop_path=0, out_features=in_features")
        self.drop_path = nn.Identity()
        self.act = act_layer(act)
        self.fc1 = Seq(nn.Conv2d(in_features, hidden_features, 1, stride=1, bias=False, padding=0),
                       nn.BatchNorm2d(hidden_features))
        self.fc2 = Seq(nn.Conv2d(hidden_features, out_features, 1, stride=1, bias=False, padding=0),
                       nn.BatchNorm2d(out_features))

    def forward_rows(self, rows):
        params, buffers = _split(self)
        return F_.run_block(F_.ffn_forward, F_.ffn_backward, params, buffers, rows, self.training)

    def forward(self, x):
        B, C, N = x.shape[0], x.shape[1], x.shape[2]
        return F_.from_rows(self.forward_rows(F_.to_rows(x)), B, N).unsqueeze(-1)


class GraphEncoder(nn.Module):
    """GraphEncoder(cfg, k=3, ...): forward(x (B, in_channels, N)) -> (B, emb_dims)  (reference :91-214).

    Extra keyword arguments (not in the reference, defaults reproduce it as shipped):
      blocks / channels : override the size table (BASELINE config 4 uses blocks=[4,4,12,4]);
      use_dilation      : False keeps dilation 1 everywhere — in the reference the block counter is never advanced
                          (graph_encoder.py:161), so `min(idx//4+1, max_dilation)` is always 1; True gives the
                          intended schedule, capped so that k*dilation fits the stage's node count.
    """

    def __init__(self, cfg, k=3, conv="mr", act="relu", norm="batch", bias=True, dropout=0.0, dilation=True,
                 epsilon=0.2, drop_path=0.1, size="t", emb_dims=1024, in_channels=3,
                 blocks=None, channels=None, use_dilation=False):
        super().__init__()
        b, c = SIZES.get(size, SIZE_DEFAULT)
        self.blocks = list(blocks) if blocks is not None else list(b)
        self.channels = list(channels) if channels is not None else list(c)
        self.k = int(k)
        self.act, self.norm, self.bias, self.drop_path, self.emb_dims = act, norm, bias, drop_path, emb_dims
        self.epsilon, self.dilation, self.dropout = epsilon, dilation, dropout
        self.num_blocks = sum(self.blocks)
        self.conv = "mr"
        N = cfg["n_mels"] * cfg["n_frames"] // (cfg["patch_bins"] * cfg["patch_frames"])
        self.num_nodes = N
        max_dilation = max(128 // self.k, 1)

        self.stem = nn.Sequential(nn.Conv2d(in_channels, self.channels[0], kernel_size=1, bias=False),
                                  nn.BatchNorm2d(self.channels[0]), nn.LeakyReLU(negative_slope=0.2))
        self.backbone = nn.ModuleList([])
        idx, n_real, n_pos = 0, N, N
        for i in range(len(self.blocks)):
            if i > 0:
                self.backbone.append(Downsample(self.channels[i - 1], self.channels[i]))
                n_pos = n_pos // 4            # the reference sizes relative_pos with N//4 per stage (:166)
                n_real = (n_real - 1) // 2 + 1
            for _ in range(self.blocks[i]):
                d = 1
                if use_dilation:
                    d = max(1, min(idx // 4 + 1, max_dilation, n_real // self.k))
                    idx += 1
                self.backbone += [Seq(
                    Grapher(self.channels[i], self.k, d, self.conv, self.act, self.norm, self.bias, False, epsilon, 1,
                            n=n_pos, drop_path=0.0, relative_pos=True),
                    FFN(in_features=self.channels[i], hidden_features=self.channels[i] * 4,
                        out_features=self.channels[i], act=act, drop_path=0.0))]
        self.backbone = Seq(*self.backbone)
        self.proj = nn.Conv2d(self.channels[-1], self.emb_dims, 1, bias=True)

    def forward_rows(self, nodes, B, N, return_nodes=False):
        """nodes (B*N, in_channels) node-major -> (B, emb_dims); with return_nodes also the pre-projection node
        matrix as (rows (B*N_last, C_last), N_last, emb)"""
        params, buffers = _split(self.stem)
        prev_chain = F_.CHAIN
        F_.CHAIN = F_.BwdChain() if (self.training and torch.is_grad_enabled()) else None
        try:
            x = F_.run_block(F_.stem_forward, F_.stem_backward, params, buffers, nodes, self.training)
            for entry in self.backbone:
                if isinstance(entry, Downsample):
                    x = entry.forward_rows(x, B, N)
                    N = (N - 1) // 2 + 1
                else:
                    x = entry[1].forward_rows(entry[0].forward_rows(x, B, N))
        finally:
            F_.CHAIN = prev_chain
        params, buffers = _split(self.proj)
        emb = F_.run_block(F_.proj_mean_forward, F_.proj_mean_backward, params, buffers, x, B, N)
        return (x, N, emb) if return_nodes else emb

    def forward(self, x):
        B, C, N = x.shape
        return self.forward_rows(F_.to_rows(x), B, N)

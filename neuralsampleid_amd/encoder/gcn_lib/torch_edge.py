"""Mirror of encoder/gcn_lib/torch_edge.py (live subset: reference :7-18, :70-103, :233-284).

One fused kernel (csrc/knn.hip) replaces F.normalize -> pairwise_distance -> topk -> [::dilation]."""
import torch
from torch import nn

from ... import functional as F_
from ... import ops


def dense_knn_matrix(x, k=16, relative_pos=None, dilation=1, normalize=False):
    """x (B, C, N, 1) -> edge_index (2, B, N, k) int64 = [neighbour idx, centre idx] (reference :70-103).
    The MI355X kernel always L2-normalises channels first; on already-normalised input (how the reference calls this
    function, :281-284) that is the identity up to rounding."""
    if relative_pos is not None:
        raise NotImplementedError("relative_pos is dead in the reference (torch_vertex.py:189) and not implemented")
    B, C, N = x.shape[0], x.shape[1], x.shape[2]
    with torch.no_grad():
        rows = ops.bcn_to_rows(x.detach().contiguous())
        nn_idx = ops.knn_graph(rows, B, N, C, k // dilation, dilation).long()
        center = torch.arange(N, device=x.device).view(1, N, 1).expand(B, N, nn_idx.shape[-1])
    return torch.stack((nn_idx, center), dim=0)


class DenseDilated(nn.Module):
    """edge_index[..., ::dilation] (reference :233-255; the stochastic branch is disabled by GraphEncoder)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        if stochastic:
            raise NotImplementedError("stochastic dilation is off on the GraFP path (graph_encoder.py:141)")
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k

    def forward(self, edge_index):
        return edge_index[:, :, :, ::self.dilation]


class DenseDilatedKnnGraph(nn.Module):
    """forward(x (B,C,N,1)) -> LongTensor (2,B,N,k): [0] neighbour ids, [1] centre ids (reference :258-284)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)

    def forward(self, x, y=None, relative_pos=None):
        if y is not None:
            raise NotImplementedError("r > 1 (pooled y) is unreachable from GraphEncoder (r=1, graph_encoder.py:171)")
        return dense_knn_matrix(x, self.k * self.dilation, relative_pos, dilation=self.dilation)

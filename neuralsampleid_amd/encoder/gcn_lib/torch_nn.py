"""Mirror of encoder/gcn_lib/torch_nn.py (reference :9-37, :52-76): layer factories and BasicConv.

The torch layers built here are PARAMETER CONTAINERS — same classes, construction order and initialisers as the
reference, so `state_dict` keys/shapes and seeded default initialisation are identical — their `forward` is never
used on the hot path: the owning block runs the HIP kernels on their tensors (functional.py)."""
import torch
from torch import nn
from torch.nn import Conv2d, Sequential as Seq


def act_layer(act, inplace=False, neg_slope=0.2, n_prelu=1):
    act = act.lower()
    if act == "relu":
        return nn.ReLU(inplace)
    if act == "leakyrelu":
        return nn.LeakyReLU(neg_slope, inplace)
    raise NotImplementedError("activation layer [%s] is not found" % act)   # prelu/gelu/hswish: not on the GraFP path


def norm_layer(norm, nc):
    norm = norm.lower()
    if norm == "batch":
        return nn.BatchNorm2d(nc, affine=True)
    raise NotImplementedError("normalization layer [%s] is not found" % norm)


class BasicConv(Seq):
    """Conv2d(1x1, groups=4) -> BatchNorm2d -> ReLU; kaiming-normal weight, zero bias (reference :52-76)."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0.0):
        if len(channels) != 2 or norm is None or str(norm).lower() != "batch" or act is None or act.lower() != "relu" \
                or drop > 0:
            raise NotImplementedError("the MI355X path implements BasicConv([cin, cout], act='relu', norm='batch')")
        m = [Conv2d(channels[0], channels[1], 1, bias=bias, groups=4), norm_layer(norm, channels[-1]), act_layer(act)]
        super().__init__(*m)
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


class _BatchedIndexSelect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x3, idx):
        from ... import ops
        ctx.save_for_backward(idx)
        ctx.n = x3.shape[2]
        return ops.batched_index_select_fwd(x3, idx)

    @staticmethod
    def backward(ctx, dout):
        from ... import ops
        (idx,) = ctx.saved_tensors
        return ops.batched_index_select_bwd(dout.contiguous(), idx, ctx.n), None


def batched_index_select(x, idx):
    """x (B, C, N, 1), idx (B, Nq, k) -> (B, C, Nq, k): features of the neighbours (reference :79-98).
    Provided for drop-in callers; MRConv2d itself never materialises this tensor (csrc/mr.hip gathers while it aggregates)."""
    if x.dim() != 4 or x.shape[3] != 1 or idx.dim() != 3 or idx.shape[0] != x.shape[0]:
        raise RuntimeError("batched_index_select expects x (B, C, N, 1) and idx (B, N, k)")
    x3 = x.reshape(x.shape[0], x.shape[1], x.shape[2]).contiguous().float()
    return _BatchedIndexSelect.apply(x3, idx.to(torch.int32).contiguous())

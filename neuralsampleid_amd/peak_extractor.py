"""Mirror of peak_extractor.py::GPUPeakExtractorv2 (reference :6-70): forward(spec (B,n_mels,n_frames)) -> (B,F,N).

One kernel per direction (csrc/misc.hip patchify_*): per-clip min-max normalise, time/frequency ramps and the
stride=kernel patch convolution + ReLU are fused; the ramps are generated in-kernel, so the reference's
batch-size-baked T/F buffers and their try/except fallback (:28-34, :56-64) have no equivalent here."""
import torch
import torch.nn as nn

from . import functional as F_


class _PatchifyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, weight, bias, pb, pf, grad_on=True):
        P = {"convs.0.weight": weight, "convs.0.bias": bias}
        need = grad_on and any(ctx.needs_input_grad)      # see functional._BlockFn.forward
        S = {} if need else None
        out = F_.patchify_forward(spec.contiguous(), P, S, pb, pf)
        ctx.S, ctx.P = S, P
        return out

    @staticmethod
    def backward(ctx, dout):
        direct = F_.DIRECT_GRADS and all(v.grad is not None for v in ctx.P.values())
        G = {k: (v.grad if direct else F_.ops.zeros(v.shape, v.device)) for k, v in ctx.P.items()}
        n_calls = len(F_.DEFERRED.calls)
        F_._IN_DIRECT_BACKWARD = direct
        try:
            F_.patchify_backward(dout.contiguous(), ctx.P, ctx.S, G)
        finally:
            F_._IN_DIRECT_BACKWARD = False
        ctx.S = None
        if direct:
            if F_.GRAD_READY_HOOK is not None:
                if len(F_.DEFERRED.calls) > n_calls:        # the launches run in the deferred phase (functional.DEFER_PATCHIFY)
                    F_.DEFERRED.note_hook(list(ctx.P.values()), len(F_.DEFERRED.items))
                else:
                    F_.GRAD_READY_HOOK(list(ctx.P.values()))
            return None, None, None, None, None, None
        return None, G["convs.0.weight"], G["convs.0.bias"], None, None, None


class GPUPeakExtractorv2(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.n_filters = cfg["n_filters"]
        self.patch_bins = cfg["patch_bins"]
        self.patch_frames = cfg["patch_frames"]
        self.convs = nn.Sequential(
            nn.Conv2d(in_channels=3, out_channels=self.n_filters, kernel_size=(self.patch_bins, self.patch_frames),
                      stride=(self.patch_bins, self.patch_frames)),
            nn.ReLU())
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward_rows(self, spec_tensor):
        """(B, n_mels, n_frames) -> node-major (B*N, F)"""
        conv = self.convs[0]
        return _PatchifyFn.apply(spec_tensor, conv.weight, conv.bias, self.patch_bins, self.patch_frames,
                                 torch.is_grad_enabled())

    def forward(self, spec_tensor):
        B, H, W = spec_tensor.shape
        N = (H // self.patch_bins) * (W // self.patch_frames)
        return F_.from_rows(self.forward_rows(spec_tensor), B, N)

"""Relative position table of encoder/gcn_lib/pos_embed.py + torch_vertex.py:165-172 (state_dict compatibility only:
the reference's Grapher.forward passes relative_pos=None, :189, so the table is never read).

The reference materialises a 2-D sin-cos embedding E (g*g, D) and takes 2*E*E^T/D.  Each (sin, cos) pair contributes
sin(a w)sin(b w) + cos(a w)cos(b w) = cos((a-b) w), so the Gram matrix has the closed form used here:
    rel[p, q] = (2/D) * sum_axis sum_d cos((pos_axis[p] - pos_axis[q]) * w_d),   w_d = 10000^(-d / (D/4)), d < D/4
evaluated in float64 like the reference, then cast to fp32, resized with bicubic interpolation to (n, n/r^2) and
negated. Agreement with the reference's values: <= 1 ulp of fp32 (tests/test_modules_cpu.py)."""
import numpy as np
import torch
import torch.nn.functional as F


def relative_pos_table(embed_dim: int, n: int, r: int = 1) -> torch.Tensor:
    """-> (1, n, n // r^2) fp32, the value of Grapher.relative_pos for `in_channels=embed_dim`, `n` nodes"""
    g = int(n ** 0.5)
    quarter = embed_dim // 4
    omega = 1.0 / 10000.0 ** (np.arange(quarter, dtype=np.float64) / quarter)
    idx = np.arange(g * g)
    col = (idx % g).astype(np.float64)          # "w goes first" in the reference's meshgrid: axis 0 = column index
    row = (idx // g).astype(np.float64)
    rel = np.zeros((g * g, g * g), dtype=np.float64)
    for pos in (col, row):
        delta = pos[:, None] - pos[None, :]
        rel += np.cos(delta[:, :, None] * omega[None, None, :]).sum(axis=-1)
    rel *= 2.0 / embed_dim
    t = torch.from_numpy(np.float32(rel)).unsqueeze(0).unsqueeze(1)
    t = F.interpolate(t, size=(n, n // (r * r)), mode="bicubic", align_corners=False)
    return -t.squeeze(1)

"""Mirror of simclr/ntxent.py::ntxent_loss (reference :5-30): one fused kernel pair instead of a 2B-iteration loop."""
from .. import functional as F_


def ntxent_loss(z_i, z_j, cfg):
    """z_i, z_j (B, d) -> 0-dim loss; reads cfg['tau']."""
    return F_.ntxent(z_i, z_j, float(cfg["tau"]))

"""Reference-format checkpoints (util.py:149-164 save_ckp / load_ckp; consumers generate.py:88-97, test_fp.py:372-383).

The reference stores {'epoch', 'loss', 'hit_rate_log', 'state_dict', 'optimizer', 'scheduler'} and, when it trained under
nn.DataParallel (train.py:117-120), every state_dict key carries a `module.` prefix that its consumers strip by hand
(generate.py:94-95, test_fp.py:381-382). The module shells here have the reference's key set, so loading is strict."""
from typing import Dict, Mapping

import torch

DATA_PARALLEL_PREFIX = "module."


def strip_data_parallel_prefix(state_dict: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """`module.encoder.stem.0.weight` -> `encoder.stem.0.weight` when (as generate.py:94 tests it) the keys are prefixed"""
    keys = list(state_dict.keys())
    if keys and all(k.startswith(DATA_PARALLEL_PREFIX) for k in keys):
        return {k[len(DATA_PARALLEL_PREFIX):]: v for k, v in state_dict.items()}
    if any(k.startswith(DATA_PARALLEL_PREFIX) for k in keys):
        raise KeyError("checkpoint mixes DataParallel-prefixed and plain keys")
    return dict(state_dict)


def _numpy_log_globals():
    """what numpy scalars / arrays pickle to: the reference logs `loss` and `hit_rate_log` entries as numpy values (train.py:150-158)"""
    import numpy as np
    out = [np.ndarray, np.dtype, np.float32, np.float64, np.int64, np.int32, np.bool_]
    core = getattr(np, "_core", None) or getattr(np, "core")
    out += [core.multiarray._reconstruct, core.multiarray.scalar]
    out += [type(np.dtype(t)) for t in ("float32", "float64", "int64", "int32", "bool")]
    # the reference pins numpy 1.26.4 (requirements.txt), whose pickles name the reconstructors `numpy.core.multiarray.*`; torch's
    # restricted unpickler matches globals by that STRING, and under numpy >= 2 the functions report `numpy._core.multiarray.*`:
    # register the legacy paths too, in torch's (callable, "full.path") form
    for fn, name in ((core.multiarray._reconstruct, "_reconstruct"), (core.multiarray.scalar, "scalar")):
        for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):
            if f"{fn.__module__}.{fn.__name__}" != f"{mod}.{name}":
                out.append((fn, f"{mod}.{name}"))
    return out


def load_reference_checkpoint(model: torch.nn.Module, path_or_dict, strict: bool = True, map_location="cpu",
                              trusted: bool = False) -> dict:
    """Load a checkpoint written by the reference's save_ckp (or a bare state_dict) into `model`; returns the checkpoint
    dict (epoch / loss / optimizer / scheduler entries untouched) so a caller can resume as train.py:131-137 does.

    A file is read with torch's restricted unpickler (`weights_only=True`) plus the numpy scalar / array reconstructors the
    reference's loss and hit-rate logs contain. `trusted=True` is the caller's statement that the file is local and theirs: only
    then does a file the restricted unpickler rejects get the unrestricted `torch.load` the reference itself uses
    (util.py:149-158, generate.py:88-97) — which executes whatever the pickle says. Never set it for a downloaded file."""
    ckpt = path_or_dict
    if not isinstance(ckpt, Mapping):
        import pickle
        try:
            with torch.serialization.safe_globals(_numpy_log_globals()):
                ckpt = torch.load(path_or_dict, map_location=map_location, weights_only=True)
        except (pickle.UnpicklingError, RuntimeError) as safe_err:
            if not trusted:
                raise pickle.UnpicklingError(
                    f"{path_or_dict}: rejected by the restricted unpickler ({safe_err}). If this is a local file you trust "
                    "(e.g. a checkpoint the reference's train.py wrote on this machine), pass trusted=True to "
                    "load_reference_checkpoint; that runs an unrestricted pickle load, which can execute arbitrary code.") from safe_err
            ckpt = torch.load(path_or_dict, map_location=map_location, weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt and isinstance(ckpt["state_dict"], Mapping) else ckpt
    model.load_state_dict(strip_data_parallel_prefix(sd), strict=strict)
    return ckpt if sd is not ckpt else {"state_dict": sd}


def save_reference_checkpoint(path, model: torch.nn.Module, epoch: int = 0, loss=None, optimizer=None, scheduler=None,
                              hit_rate_log=None) -> None:
    """the dict layout of train.py:150-158. hit_rate_log defaults to the empty list the reference always writes. A checkpoint that
    the reference's util.load_ckp is to RESUME from needs `optimizer` and `scheduler` (it calls load_state_dict on both, util.py:152-153);
    without them the file serves inference consumers (generate.py:88-97, test_fp.py:372-383), which read `state_dict` only."""
    torch.save({"epoch": epoch, "loss": loss, "hit_rate_log": [] if hit_rate_log is None else hit_rate_log,
                "state_dict": model.state_dict(),
                "optimizer": None if optimizer is None else optimizer.state_dict(),
                "scheduler": None if scheduler is None else scheduler.state_dict()}, path)

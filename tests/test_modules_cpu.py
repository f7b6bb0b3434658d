"""CPU: the module shells keep the reference's state_dict layout and seeded default initialisation."""
import json
import os

import torch

from conftest import GOLDEN
from synth import GRAFP_CFG


def build(k=3, **kw):
    from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
    from neuralsampleid_amd.simclr.simclr import SimCLR
    return SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG, in_channels=GRAFP_CFG["n_filters"], k=k, size="t", **kw))


def test_state_dict_keys_and_shapes_match_reference():
    with open(os.path.join(GOLDEN, "state_shapes.json")) as f:
        ref = json.load(f)
    sd = build().state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(list(sd[k].shape) == ref[k] for k in ref)
    assert sd["encoder.stem.1.num_batches_tracked"].dtype == torch.int64


def test_default_init_matches_reference_under_seed():
    with open(os.path.join(GOLDEN, "init_seed42_checksums.json")) as f:
        chk = json.load(f)
    torch.manual_seed(42)
    sd = build().state_dict()
    for k, (s, n) in chk.items():
        v = sd[k].double()
        assert abs(float(v.sum()) - s) <= 1e-9 * max(1.0, abs(s)) and abs(float(v.norm()) - n) <= 1e-9 * max(1.0, n), k


def test_trainable_parameter_count_and_frozen_relative_pos():
    m = build()
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 18366856
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert len(frozen) == 12 and all(n.endswith("relative_pos") for n in frozen)
    assert sum(p.numel() for n, p in m.named_parameters() if not p.requires_grad) == 140832


def test_deep_config_plan():
    """BASELINE config 4: blocks x2, k=18, intended dilation schedule capped by the stage's node count"""
    m = build(k=18, blocks=[4, 4, 12, 4], use_dilation=True)
    ds = [blk[0].graph_conv.d for blk in m.encoder.backbone if not hasattr(blk, "conv")]
    ns = [256] * 4 + [128] * 4 + [64] * 12 + [32] * 4
    assert len(ds) == 24 and ds[0] == 1 and ds[4] == 2 and all(18 * d <= n for d, n in zip(ds, ns))
    from oracle import ref_torch as R
    plan = R.encoder_plan("t", 18, blocks=[4, 4, 12, 4], use_dilation=True)
    assert [e[3] for e in plan if e[0] == "block"] == ds


def test_relative_pos_matches_reference(golden):
    """the dead relative_pos buffers carry the reference's values (<= 1 fp32 ulp)"""
    g = golden("relative_pos_t")
    sd = build().state_dict()
    for key in g:
        ref = g.t(key)
        got = sd["encoder." + key]
        assert got.shape == ref.shape and not got.requires_grad
        assert float((got - ref).abs().max()) <= 2.4e-7 * max(1.0, float(ref.abs().max())), key


def test_load_reference_checkpoint_strips_the_dataparallel_prefix(tmp_path):
    """generate.py:94-95 / test_fp.py:381-382: checkpoints of a DataParallel run carry `module.` on every key; the
    reference's checkpoint dict is util.py:160-164's {'epoch', 'loss', ..., 'state_dict', 'optimizer', 'scheduler'}"""
    import pytest
    from neuralsampleid_amd.checkpoint import (load_reference_checkpoint, save_reference_checkpoint,
                                               strip_data_parallel_prefix)
    torch.manual_seed(3)
    src = build()
    for b in src.buffers():                                     # make the BN buffers distinguishable from a fresh model
        if b.dtype.is_floating_point:
            b.add_(torch.rand_like(b))
    sd = src.state_dict()
    ck = {"epoch": 7, "loss": 1.5, "hit_rate_log": [], "optimizer": {}, "scheduler": {},
          "state_dict": {"module." + k: v for k, v in sd.items()}}           # what nn.DataParallel(model).state_dict() is
    path = tmp_path / "model_tc_35_best.pth"
    torch.save(ck, path)
    torch.manual_seed(4)
    dst = build()
    out = load_reference_checkpoint(dst, str(path))                           # strict=True: all 443 keys must match
    assert out["epoch"] == 7
    got = dst.state_dict()
    assert list(got.keys()) == list(sd.keys()) and all(torch.equal(got[k], sd[k]) for k in sd)
    # un-prefixed checkpoint and a bare state_dict load the same way
    torch.manual_seed(5)
    dst2 = build()
    load_reference_checkpoint(dst2, {"state_dict": sd})
    load_reference_checkpoint(dst2, sd)
    assert all(torch.equal(dst2.state_dict()[k], sd[k]) for k in sd)
    with pytest.raises(KeyError):
        strip_data_parallel_prefix({"module.a": torch.zeros(1), "b": torch.zeros(1)})
    with pytest.raises(RuntimeError):                                          # strict: a missing key is an error
        load_reference_checkpoint(dst2, {k: v for k, v in sd.items() if k != "encoder.proj.bias"})
    save_reference_checkpoint(str(tmp_path / "out.pth"), src, epoch=3, loss=0.5)
    back = torch.load(tmp_path / "out.pth")
    assert sorted(back) == ["epoch", "hit_rate_log", "loss", "optimizer", "scheduler", "state_dict"]


def test_batched_index_select_symbol_exists_with_the_reference_signature():
    import inspect
    from neuralsampleid_amd.encoder.gcn_lib.torch_nn import batched_index_select
    assert list(inspect.signature(batched_index_select).parameters) == ["x", "idx"]      # torch_nn.py:79


def test_checkpoint_loading_is_restricted_unless_the_caller_says_trusted(tmp_path):
    """a checkpoint is read with torch's restricted unpickler (+ the numpy scalars the reference's logs hold, train.py:150-158);
    anything it rejects is loaded with an unrestricted pickle only when the caller passes trusted=True."""
    import pickle
    import numpy as np
    import pytest
    from neuralsampleid_amd.checkpoint import load_reference_checkpoint
    torch.manual_seed(6)
    src = build()
    sd = src.state_dict()
    # (a) numpy values in the logs, as the reference writes them: accepted by the restricted load
    ck = {"epoch": 2, "loss": np.float64(0.75), "hit_rate_log": [np.float32(0.5), np.array([1.0, 2.0])], "state_dict": sd,
          "optimizer": None, "scheduler": None}
    p1 = tmp_path / "np_logs.pth"
    torch.save(ck, p1)
    dst = build()
    out = load_reference_checkpoint(dst, str(p1))
    assert float(out["loss"]) == 0.75 and all(torch.equal(dst.state_dict()[k], sd[k]) for k in sd)

    # (b) a pickle that names an arbitrary global: refused with a hint, loaded only with trusted=True
    class Marker:
        pass
    globals()["Marker"] = Marker                      # picklable by reference
    Marker.__module__, Marker.__qualname__ = __name__, "Marker"
    p2 = tmp_path / "arbitrary.pth"
    torch.save({"state_dict": sd, "extra": Marker()}, p2)
    with pytest.raises(pickle.UnpicklingError, match="trusted=True"):
        load_reference_checkpoint(build(), str(p2))
    dst3 = build()
    load_reference_checkpoint(dst3, str(p2), trusted=True)
    assert all(torch.equal(dst3.state_dict()[k], sd[k]) for k in sd)


def test_checkpoint_with_numpy1_pickle_paths_loads_restricted(tmp_path):
    """the reference pins numpy 1.26.4, whose pickles name `numpy.core.multiarray.scalar` / `._reconstruct`; under numpy 2 the
    installed functions report `numpy._core.multiarray.*` and torch's restricted unpickler matches globals by string. A reference-era
    checkpoint (data.pkl rewritten to the numpy-1.x module path) must load WITHOUT trusted=True."""
    import zipfile
    import numpy as np
    from neuralsampleid_amd.checkpoint import load_reference_checkpoint
    torch.manual_seed(7)
    src = build()
    sd = src.state_dict()
    ck = {"epoch": 1, "loss": np.float64(1.25), "hit_rate_log": [np.float32(0.25), np.array([3.0, 4.0])], "state_dict": sd,
          "optimizer": None, "scheduler": None}
    p_new, p_old = tmp_path / "np2.pth", tmp_path / "np1.pth"
    torch.save(ck, p_new)                                  # protocol 2: globals are newline-terminated text (`cmodule\nname\n`)
    n_rewritten = 0
    with zipfile.ZipFile(p_new) as zin, zipfile.ZipFile(p_old, "w", zipfile.ZIP_STORED) as zout:
        for item in zin.infolist():
            data = zin.read(item.filename)
            if item.filename.endswith("data.pkl"):
                n_rewritten = data.count(b"cnumpy._core.multiarray\n")
                data = data.replace(b"cnumpy._core.multiarray\n", b"cnumpy.core.multiarray\n")
            zout.writestr(item, data)
    if n_rewritten == 0:                                   # numpy 1.x installed: the file already names the legacy path
        with zipfile.ZipFile(p_new) as zin:
            assert any(b"cnumpy.core.multiarray\n" in zin.read(i.filename) for i in zin.infolist() if i.filename.endswith("data.pkl"))
    dst = build()
    out = load_reference_checkpoint(dst, str(p_old))
    assert float(out["loss"]) == 1.25 and float(out["hit_rate_log"][0]) == 0.25 and list(out["hit_rate_log"][1]) == [3.0, 4.0]
    assert all(torch.equal(dst.state_dict()[k], sd[k]) for k in sd)

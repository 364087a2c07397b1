"""tests/golden/ckpt_r256.npz -> a checkpoint file in the reference's own format (train.py:308-322).

The fixture holds the tensors of the file the reference's unmodified train() saved after iteration 1 (tests/golden/make_golden.py
``ckpt``): every state-dict entry of the 11 networks in the reference's key order, the Adam state of the 3 optimisers (``step`` /
``exp_avg`` / ``exp_avg_sq`` per parameter index, ``param_groups`` as JSON) and the args as JSON.  The reference's
ImageLevelDiscriminator cannot be narrowed (models.py:336-341), so its 13 tensors above 65536 elements and their Adam state (288 MB)
are listed under ``ck.omitted`` with shape, f64 sum / abs-sum and 8 seeded projections instead of being stored; ``fill(key, shape)``
supplies them (the GPU test: from its own iteration 1; the CPU format test: zeros)."""
import argparse
import collections
import json

import numpy as np
import torch

from conftest import Golden, sketch


def omitted_entries(g: Golden):
    """[(fixture key, shape, f64 sum, f64 abs-sum, sketch)] of the tensors the fixture does not store."""
    return [tuple(o) for o in g.json("ck.omitted")]


def build_reference_checkpoint(g: Golden, fill):
    """The dict the reference's train() handed to torch.save (train.py:313-319), rebuilt from the fixture's arrays."""
    omitted = {o[0]: o[1] for o in g.json("ck.omitted")}

    def tensor(key):
        if key in omitted:
            t = fill(key, tuple(omitted[key]))
            assert tuple(t.shape) == tuple(omitted[key]) and t.dtype == torch.float32, key
            return t.detach().cpu().contiguous()
        return g.t(key)

    trainer = collections.OrderedDict()
    for name in g.json("ck.trainer_keys"):
        if name.endswith("_optim"):
            state = {}
            for idx, keys in g.json(f"ck.{name}.state_keys"):
                state[idx] = {k: tensor(f"ck.{name}.state.{idx}.{k}") for k in keys}
            groups = g.json(f"ck.{name}.param_groups")
            for grp in groups:
                grp["betas"] = tuple(grp["betas"])
            trainer[name] = {"state": state, "param_groups": groups}
        else:
            trainer[name] = collections.OrderedDict((k, tensor(f"ck.{name}/{k}")) for k in g.json(f"ck.{name}.keys"))
    a = g.json("ck.args")
    if isinstance(a.get("channel_multiplier"), str):          # the generator's int-like 1/den (make_golden.Shrink)
        num, den = a["channel_multiplier"].split("/")
        a["channel_multiplier"] = float(num) / float(den)
    a["blur_kernel"] = tuple(a["blur_kernel"])
    return {"iter_idx": int(g.t("ck.iter_idx")), "N": int(g.t("ck.N")), "trainer": trainer, "args": argparse.Namespace(**a)}


def check_against_omitted(g: Golden, key: str, t: torch.Tensor, index_of, rel_sum: float, rel_dir: float):
    """A tensor supplied for an omitted entry against what the fixture kept of the reference's: abs-sum and direction sketch."""
    for k, shape, s_ref, a_ref, sk_ref in omitted_entries(g):
        if k != key:
            continue
        t64 = t.detach().double().cpu()
        assert tuple(t.shape) == tuple(shape), (key, t.shape, shape)
        a = float(t64.abs().sum())
        assert abs(a - a_ref) <= rel_sum * a_ref, (key, a, a_ref)
        sk = torch.tensor(sketch(t, index_of(key)), dtype=torch.float64)
        err = float((sk - torch.tensor(sk_ref, dtype=torch.float64)).norm() / (8 ** 0.5 * float(t64.norm()) + 1e-300))
        assert err <= rel_dir, (key, err)
        return
    raise KeyError(key)


def sketch_index(g: Golden):
    """The seed index make_golden.py used for each omitted tensor (7000 + parameter index for Adam state, 8000 + state-dict position)."""
    keys = g.json("ck.Dreal.keys")

    def index_of(key: str) -> int:
        if key.startswith("ck.d_optim.state."):
            return 7000 + int(key.split(".")[3])
        return 8000 + keys.index(key[len("ck.Dreal/"):])
    return index_of

"""Run REAL method bodies of the reference trainer without importing it (build container only).

vlnce_baselines/ss_trainer_ETP.py imports habitat, lmdb, gym ... at module level, but the two input-assembly methods on
the planner's boundary — RLTrainer._vp_feature_variable (:308-342) and RLTrainer._nav_gmap_variable (:344-417) — only
need numpy, torch, pad_sequence and the reference's own common/ops.py.  This module cuts their FunctionDefs out of the
source with `ast`, compiles them unchanged into a namespace holding exactly those names, and calls them with a stand-in
`self`.  Shims for running on this CPU-only, numpy-2 container (no effect on the arithmetic): `np.bool` (removed from
numpy) is aliased to `bool` and `Tensor.cuda()` is the identity while the functions run.
"""
from __future__ import annotations

import ast
import importlib.util
import types

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

TRAINER = "/root/reference/vlnce_baselines/ss_trainer_ETP.py"
OPS = "/root/reference/vlnce_baselines/common/ops.py"


def _load_ops():
    from oracle import ref_harness as rh
    rh._import_vilmodel()                       # registers the stub packages; vilmodel_cmt imports common.ops itself
    import importlib
    return importlib.import_module("vlnce_baselines.common.ops")


def extract(names, extra_globals=None):
    src = open(TRAINER).read()
    tree = ast.parse(src)
    fns = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            fns[node.name] = node
    missing = set(names) - set(fns)
    assert not missing, missing
    ops = _load_ops()
    ns = {"np": np, "torch": torch, "pad_sequence": pad_sequence, "pad_tensors_wgrad": ops.pad_tensors_wgrad,
          "gen_seq_masks": ops.gen_seq_masks}
    ns.update(extra_globals or {})
    mod = ast.Module(body=[fns[n] for n in names], type_ignores=[])
    exec(compile(mod, TRAINER, "exec"), ns)
    return {n: ns[n] for n in names}


class shims:
    """Context manager: np.bool alias + Tensor.cuda identity (CPU-only container)."""

    def __enter__(self):
        self._had = hasattr(np, "bool")
        if not self._had:
            np.bool = bool
        self._cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda = self._cuda
        if not self._had:
            del np.bool
        return False


def fake_self(**attrs):
    s = types.SimpleNamespace(**attrs)
    return s

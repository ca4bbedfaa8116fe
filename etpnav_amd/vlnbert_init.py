"""Drop-in for vlnce_baselines/models/etp/vlnbert_init.py: ``get_vlnbert_models(config)`` builds the MI355X-native
planner with the hyper-parameters the reference hard-codes (vlnbert_init.py:32-59) and loads a pre-training
checkpoint with the reference's key remapping (:20-30)."""
from __future__ import annotations

import torch

from .planner import GlocalTextPathNavCMT, default_config


def remap_checkpoint_keys(ckpt_weights):
    """vlnbert_init.py:22-30: strip a leading 'module.', and lose the 'bert.' prefix that HF's from_pretrained would
    strip (the reference *adds* 'bert.' to sap_head keys so that stripping 'bert.' from everything lines them up)."""
    out = {}
    for k, v in ckpt_weights.items():
        if k.startswith("module."):
            k = k[7:]
        if k.startswith("bert."):
            k = k[5:]
        out[k] = v
    return out


def get_vlnbert_models(config=None, dtype: torch.dtype = torch.bfloat16, device=None):
    """config: the habitat ``MODEL`` node (or any object) with pretrained_path, task_type in {'r2r','rxr'},
    use_depth_embedding, use_sprels, fix_lang_embedding, fix_pano_embedding — as read at vlnbert_init.py:20-54."""
    task_type = getattr(config, "task_type", "r2r")
    vis_config = default_config(
        task_type,
        use_depth_embedding=getattr(config, "use_depth_embedding", True),
        graph_sprels=getattr(config, "use_sprels", True),
        fix_lang_embedding=getattr(config, "fix_lang_embedding", False),
        fix_pano_embedding=getattr(config, "fix_pano_embedding", False),
    )
    vis_config.update_lang_bert = not vis_config.fix_lang_embedding
    model = GlocalTextPathNavCMT(vis_config, dtype=dtype, device=device)
    path = getattr(config, "pretrained_path", None)
    if path is not None:
        state = remap_checkpoint_keys(torch.load(path, map_location="cpu"))
        own = set(model.state_dict().keys())
        # pre-training checkpoints carry extra heads (mlm_head, lang_* of the x-layers): load what the planner has,
        # as HF from_pretrained(strict=False) does for the reference
        model.load_state_dict({k: v for k, v in state.items() if k in own}, strict=False)
    return model

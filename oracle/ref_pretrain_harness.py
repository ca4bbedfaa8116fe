"""Build the REAL pre-training model (pretrain_src/pretrain_src/model/pretrain_cmt.py:50 GlocalTextPathCMTPreTraining) in
the build container and run its SAP / MLM tasks on oracle-shaped batches.  Test infrastructure (golden generation) only.

Stubs: a package shell over pretrain_src/pretrain_src/model so its relative imports resolve;
transformers-5 lacks the 4.12 helpers the constructor calls, so `init_weights` is the BERT-style initialiser of
oracle/ref_harness.py and `_tie_or_clone_weights` shares the decoder weight with the word embeddings (the non-torchscript
branch of the 4.12 method).  The model's own code is untouched.
"""
from __future__ import annotations

import importlib
import sys
import types

import torch

from oracle import ref_harness as rh

PT = "/root/reference/pretrain_src/pretrain_src"
CFG = "/root/reference/pretrain_src/run_pt/r2r_model_config_dep.json"
_pc = None


def _import():
    global _pc
    if _pc is not None:
        return _pc
    rh._import_vilmodel()                                       # installs the init_weights stand-in
    m = types.ModuleType("etp_ref_ptmodel")
    m.__path__ = [PT + "/model"]
    sys.modules["etp_ref_ptmodel"] = m
    from transformers import BertPreTrainedModel

    def _tie(self, out_emb, in_emb):
        out_emb.weight = in_emb.weight
    BertPreTrainedModel._tie_or_clone_weights = _tie
    _pc = importlib.import_module("etp_ref_ptmodel.pretrain_cmt")
    return _pc


def build_pretrain_model(cfg, params, tasks=("mlm", "sap")):
    """cfg: oracle PlannerConfig; params: oracle parameter dict (fine-tune names + lang_* / mlm_head entries)."""
    pc = _import()
    from transformers import PretrainedConfig
    vc = PretrainedConfig.from_json_file(CFG)
    vc.pretrain_tasks = list(tasks)
    for k in ("vocab_size", "type_vocab_size", "max_position_embeddings", "hidden_size", "num_attention_heads",
              "intermediate_size", "layer_norm_eps", "max_action_steps", "image_feat_size", "depth_feat_size",
              "angle_feat_size", "num_l_layers", "num_pano_layers", "num_x_layers", "graph_sprels"):
        setattr(vc, k, getattr(cfg, k))
    vc.use_lang2visn_attn = True
    model = pc.GlocalTextPathCMTPreTraining(vc)
    sd = {}
    for k, v in params.items():
        sd[k if k.startswith(("mlm_head.", "global_sap_head.")) else "bert." + k] = v.float()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    missing = [m for m in missing if m != "mlm_head.predictions.decoder.weight"]       # tied to the word embeddings
    assert not missing, missing
    model.tie_weights()
    model.eval()
    return model


def to_ref_batch(batch):
    """Oracle SAP/MLM batch (etpnav_amd.synthetic.make_sap_batch layout) -> the reference's batch dict (tasks.py collate)."""
    tr = batch["traj"]
    return {
        "txt_ids": batch["txt_ids"], "txt_lens": batch["txt_masks"].sum(1),
        "traj_view_img_fts": batch["rgb_fts"], "traj_view_dep_fts": batch["dep_fts"], "traj_obj_img_fts": None,
        "traj_loc_fts": batch["loc_fts"], "traj_nav_types": batch["nav_types"], "traj_step_lens": list(tr["traj_step_lens"]),
        "traj_vp_view_lens": batch["view_lens"], "traj_vp_obj_lens": None, "traj_vpids": tr["traj_vpids"],
        "traj_cand_vpids": tr["traj_cand_vpids"], "gmap_lens": batch["gmap_masks"].sum(1),
        "gmap_step_ids": batch["gmap_step_ids"], "gmap_pos_fts": batch["gmap_pos_fts"],
        "gmap_pair_dists": batch["gmap_pair_dists"], "gmap_vpids": tr["gmap_vpids"],
        "gmap_visited_masks": batch["gmap_visited_masks"], "global_act_labels": batch["labels"], "local_act_labels": None,
        "txt_labels": batch.get("txt_labels"),
    }

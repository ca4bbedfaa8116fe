"""Import the *real* reference planner (/root/reference) on CPU.  TEST INFRASTRUCTURE ONLY.

Only usable in the build container (the GPU box has no /root/reference); used
by ``oracle/make_golden.py`` to generate the fixtures in ``tests/golden/`` and by
``tests/test_oracle_vs_reference.py`` (skipped when the reference is absent).
Follows SURVEY.md Appendix D: stub packages so ``vilmodel_cmt.py`` imports without
habitat, replace ``BertPreTrainedModel.init_weights`` (transformers-5 drift).  No
reference file is copied or modified.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("ETPNAV_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF, "vlnce_baselines/models/etp/vilmodel_cmt.py"))


_vm = None


def _import_vilmodel():
    global _vm
    if _vm is not None:
        return _vm
    for name, sub in [("vlnce_baselines", ""), ("vlnce_baselines.common", "/common"),
                      ("vlnce_baselines.models", "/models"), ("vlnce_baselines.models.etp", "/models/etp")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [REF + "/vlnce_baselines" + sub]
            sys.modules[name] = m
    from transformers import BertPreTrainedModel

    def _bert_init(self):
        def f(m):
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(0, 0.02)
            if isinstance(m, nn.LayerNorm):
                m.bias.data.zero_(); m.weight.data.fill_(1.0)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()
        self.apply(f)

    BertPreTrainedModel.init_weights = _bert_init
    _vm = importlib.import_module("vlnce_baselines.models.etp.vilmodel_cmt")
    return _vm


def build_reference_model(cfg, params=None):
    """Build the reference GlocalTextPathNavCMT with the hyper-parameters of
    vlnbert_init.py:32-59 taken from an oracle PlannerConfig, optionally loading an
    oracle parameter dict (same names)."""
    vm = _import_vilmodel()
    from transformers import PretrainedConfig
    name = "bert-base-uncased" if cfg.vocab_size == 30522 else "xlm-roberta-base"
    vc = PretrainedConfig.from_pretrained(os.path.join(REF, "bert_config", name))
    vc.vocab_size = cfg.vocab_size
    vc.type_vocab_size = cfg.type_vocab_size
    vc.max_position_embeddings = cfg.max_position_embeddings
    vc.hidden_size = cfg.hidden_size
    vc.num_attention_heads = cfg.num_attention_heads
    vc.intermediate_size = cfg.intermediate_size
    vc.layer_norm_eps = cfg.layer_norm_eps
    vc.max_action_steps = cfg.max_action_steps
    vc.image_feat_size = cfg.image_feat_size
    vc.use_depth_embedding = cfg.use_depth_embedding
    vc.depth_feat_size = cfg.depth_feat_size
    vc.angle_feat_size = cfg.angle_feat_size
    vc.num_l_layers = cfg.num_l_layers
    vc.num_pano_layers = cfg.num_pano_layers
    vc.num_x_layers = cfg.num_x_layers
    vc.graph_sprels = cfg.graph_sprels
    vc.glocal_fuse = "global"
    vc.fix_lang_embedding = bool(getattr(cfg, "fix_lang_embedding", False))
    vc.fix_pano_embedding = bool(getattr(cfg, "fix_pano_embedding", False))
    vc.update_lang_bert = not vc.fix_lang_embedding             # vlnbert_init.py:54
    vc.output_attentions = True
    vc.pred_head_dropout_prob = 0.1
    vc.use_lang2visn_attn = False
    model = vm.GlocalTextPathNavCMT(vc)
    if params is not None:
        missing, unexpected = model.load_state_dict({k: v.float() for k, v in params.items()}, strict=True)
    model.eval()
    return model


def reference_step(model, batch):
    """The §8d unit of work on the reference model (eval mode), with autograd."""
    import torch.nn.functional as F
    from oracle.planner_oracle import assemble_gmap_img_fts
    rgb = batch["rgb_fts"].detach().clone().requires_grad_(True)
    for p in model.parameters():
        p.grad = None
    txt = model.forward_txt(batch["txt_ids"], batch["txt_masks"])
    pano, pmask = model.forward_panorama(rgb, batch["dep_fts"], batch["loc_fts"],
                                         batch["nav_types"], batch["view_lens"])
    G = batch["gmap_step_ids"].shape[1]
    gimg = assemble_gmap_img_fts(pano, pmask, batch["view_lens"], G)
    outs = model.forward_navigation(txt, batch["txt_masks"], None, batch["gmap_step_ids"], gimg,
                                    batch["gmap_pos_fts"], batch["gmap_masks"],
                                    batch["gmap_visited_masks"], batch["gmap_pair_dists"])
    B = batch["txt_ids"].shape[0]
    loss = F.cross_entropy(outs["global_logits"], batch["labels"], reduction="sum",
                           ignore_index=-100) / B
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in model.named_parameters()}
    grads["__input__.rgb_fts"] = rgb.grad.detach().clone()
    return ({"txt_embeds": txt.detach(), "pano_embeds": pano.detach(), "pano_masks": pmask,
             "gmap_img_fts": gimg.detach(), "gmap_embeds": outs["gmap_embeds"].detach(),
             "global_logits": outs["global_logits"].detach(), "loss": loss.detach()}, grads)

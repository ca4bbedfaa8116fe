"""Habitat-free mirror of the planner-facing part of vlnce_baselines/models/Policy_ViewSelection_ETP.py.

``ETP.forward(mode=...)`` keeps the reference's keyword names and dispatch (Policy_ViewSelection_ETP.py:157-170,
:344-358) for the three planner modes; the waypoint / perception modes depend on habitat, CLIP and DD-PPO encoders and
stay with the reference (out of scope).  ``PolicyViewSelectionETP`` mirrors ILPolicy (models/policy.py:
12-19): it just holds ``.net``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .vlnbert_init import get_vlnbert_models


class ETP(nn.Module):
    def __init__(self, model_config=None, dtype: torch.dtype = torch.bfloat16, device=None, fuse_drop_env: bool = True):
        super().__init__()
        self.vln_bert = get_vlnbert_models(config=model_config, dtype=dtype, device=device)
        self.drop_env = nn.Dropout(p=0.4)          # Policy_ViewSelection_ETP.py:102
        # fused: the p=0.4 feature dropout rides in forward_panorama's operand cast (and its mask is recomputed for the
        # img_linear weight gradient and d rgb_fts) instead of a separate elementwise pass over [B,V,F] + a saved mask
        self.fuse_drop_env = fuse_drop_env

    def forward(self, mode=None, txt_ids=None, txt_masks=None, txt_embeds=None, waypoint_predictor=None,
                observations=None, in_train=True, rgb_fts=None, dep_fts=None, loc_fts=None, nav_types=None,
                view_lens=None, gmap_vp_ids=None, gmap_step_ids=None, gmap_img_fts=None, gmap_pos_fts=None,
                gmap_masks=None, gmap_visited_masks=None, gmap_pair_dists=None):
        if mode == "language":
            return self.vln_bert.forward_txt(txt_ids, txt_masks)
        if mode == "panorama":
            if self.fuse_drop_env:
                self.vln_bert.drop_env_prob = self.drop_env.p if self.training else 0.0
            else:
                self.vln_bert.drop_env_prob = 0.0
                rgb_fts = self.drop_env(rgb_fts)   # :345 (identity in eval())
            return self.vln_bert.forward_panorama(rgb_fts, dep_fts, loc_fts, nav_types, view_lens)
        if mode == "navigation":
            return self.vln_bert.forward_navigation(txt_embeds, txt_masks, gmap_vp_ids, gmap_step_ids, gmap_img_fts,
                                                    gmap_pos_fts, gmap_masks, gmap_visited_masks, gmap_pair_dists)
        if mode == "waypoint":
            raise NotImplementedError("mode='waypoint' (CLIP + DD-PPO encoders + waypoint predictor) stays in the reference; "
                                      "this package replaces the planner modes only")
        raise NotImplementedError(mode)


class PolicyViewSelectionETP(nn.Module):
    """Holds ``.net`` like ILPolicy; ``from_config`` keeps the reference signature (observation/action spaces unused)."""

    def __init__(self, observation_space=None, action_space=None, model_config=None, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.net = ETP(model_config=model_config, dtype=dtype, device=device)

    @classmethod
    def from_config(cls, config, observation_space=None, action_space=None, **kw):
        model_config = getattr(config, "MODEL", config)
        return cls(observation_space=observation_space, action_space=action_space, model_config=model_config, **kw)

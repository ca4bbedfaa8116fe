"""Per-tensor bf16 error of one golden fixture against the oracle (full tensors) and the fixture samples: which tensors sit closest
to the bounds of tests/golden_util.py, for the library selected by ETP_LIB (same-box A/B of a numerics-affecting change).

    python tools/parity_probe.py [--case c1_single_episode] [--top 8]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from oracle import planner_oracle as po          # checker only (a tool, not the product)
from tests.golden_util import load_case, sample_idx
from etpnav_amd.planner import GlocalTextPathNavCMT
from etpnav_amd.step import PlannerStep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="c1_single_episode")
    ap.add_argument("--top", type=int, default=8)
    ap.add_argument("--overlap", type=int, default=1, help="0: the whole step on one stream")
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    z, cfg, batch = load_case(a.case)
    P = po.init_params(cfg, seed=0)
    outs, ref = po.step_with_grads(P, cfg, batch)
    m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32, device="cuda")
    m.load_state_dict(P, strict=True); m.eval()
    step = PlannerStep(m, batch, overlap=bool(a.overlap))
    step.run_eager(); torch.cuda.synchronize()
    pm = outs["pano_masks"]
    for k, t in (("txt_embeds", step.txt), ("pano_embeds", step.pano), ("gmap_embeds", step.gemb), ("d_txt", None)):
        if t is None:
            continue
        got, want = t.float().cpu(), outs[k]
        if k == "pano_embeds":
            got, want = got[pm], want[pm]
        print(f"# output {k}: max abs err {float((got - want).abs().max()):.3e}, relative L2 {float((got - want).norm() / want.norm()):.4f}")
    fin = torch.isfinite(outs["global_logits"])
    print(f"# output logits: max abs err {float((step.logits.float().cpu()[fin] - outs['global_logits'][fin]).abs().max()):.3e}; loss {step.loss.item():.6f} vs {outs['loss'].item():.6f}")
    rows = []
    for k, p in m.named_parameters():
        g = p.grad.detach().double().cpu().reshape(-1); r = ref[k].double().reshape(-1)
        idx = torch.from_numpy(sample_idx(g.numel()))
        smp = torch.from_numpy(z[f"gsm.{k}"]).double()
        amax = float(z[f"gfp.{k}"][1])
        if amax < 1e-6:            # zero in exact arithmetic (key biases: softmax shift invariance)
            continue
        rows.append((float((g[idx] - smp).abs().max()) / max(amax, 1e-12), float((g - r).norm() / max(float(r.norm()), 1e-12)),
                     float((g - r).sum() / max(float((g - r).abs().sum()), 1e-30)), amax, k))
    print(f"# {a.case}, library {os.environ.get('ETP_LIB', 'default')}: sample err / abs-max, full-tensor relative L2, signed-error fraction, abs-max")
    for r in sorted(rows, reverse=True)[:a.top]:
        print(f"  {r[0]:.4f}  {r[1]:.4f}  {r[2]:+.3f}  {r[3]:.3e}  {r[4]}")
    # the backward's order of appearance on the node side, then the text side: where does an error first show?
    chain = ["global_sap_head.net.4.weight", "global_sap_head.net.2.weight", "global_sap_head.net.0.weight"]
    for l in (3, 2, 1, 0):
        x = f"global_encoder.encoder.x_layers.{l}."
        chain += [x + "visn_output.LayerNorm.weight", x + "visn_output.dense.weight", x + "visn_inter.dense.weight", x + "visn_self_att.output.dense.weight",
                  x + "visn_self_att.self.value.weight", x + "visual_attention.output.dense.weight", x + "visual_attention.att.value.weight",
                  x + "visual_attention.att.query.weight"]
    chain += ["global_encoder.gmap_pos_embeddings.0.weight", "img_embeddings.pano_encoder.layers.1.linear2.weight", "img_embeddings.pano_encoder.layers.1.linear1.weight",
              "img_embeddings.img_linear.weight", "lang_encoder.layer.8.output.dense.weight", "lang_encoder.layer.8.intermediate.dense.weight",
              "lang_encoder.layer.8.attention.self.value.weight", "lang_encoder.layer.0.output.dense.weight", "embeddings.LayerNorm.bias"]
    by_name = {r[4]: r for r in rows}
    for k in chain:
        if k in by_name:
            r = by_name[k]
            print(f"# chain  {r[0]:.4f}  {r[1]:.4f}  {r[2]:+.3f}  {r[3]:.3e}  {r[4]}")
    by_l2 = sorted(rows, key=lambda r: -r[1])[:a.top]
    print("# by full-tensor relative L2")
    for r in by_l2:
        print(f"  {r[0]:.4f}  {r[1]:.4f}  {r[2]:+.3f}  {r[3]:.3e}  {r[4]}")


if __name__ == "__main__":
    main()

"""round 5, last GPU call: WHAT makes pano_embed_bwd nondeterministic when it shares CUs?

Run with ETP_LIB=etpnav_amd/build/libetp_panoexpt.so (tools/experiments/r05_pano_bwd_isolation_build.py).  Ten repetitions of the
step on identical inputs per mode; reported per mode: the largest deviation from the per-element median (relative to the tensor's
abs-max) over (a) the gradients pano_embed_bwd produces, (b) the panorama encoder's gradients (upstream of it), (c) all other
gradients -- and which BYTES of the panorama backward's workspace differ between repetitions (the kernel's input dy and its outputs
da / dd live there: inputs identical + outputs different = the kernel itself; inputs different = its producers / a dependency).

  A  three streams, 12 KB LDS                      (the form the race screen rejects)
  B  three streams, 12 KB, device drain BEFORE the three launches       (earlier work cannot overlap them; later work can)
  C  three streams, 12 KB, drain before AND after                       (the launches run alone)
  D  ONE stream, 12 KB, unrelated matrix products (torch / rocBLAS) on a foreign stream    (co-resident wavefronts that touch none of
                                                                                             the planner's memory)
  E  ONE stream, 12 KB, nothing else                                    (control)
  F  three streams, 160 KB (the product)                                (control)
"""
import os
import sys
import time

t_start = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from etpnav_amd.planner import GlocalTextPathNavCMT, default_config  # noqa: E402
from etpnav_amd.step import PlannerStep  # noqa: E402
from etpnav_amd.synthetic import make_batch  # noqa: E402

print(f"imports {time.time() - t_start:.1f} s", flush=True)
RUNS = int(os.environ.get("RUNS", "10"))
WLS = os.environ.get("WL", "c5,c2").split(",")
w = dict(bench.WORKLOADS[WLS[0]])
cfg = default_config(w["task"], image_feat_size=w["image_feat_size"])
model = GlocalTextPathNavCMT(cfg, dtype=torch.bfloat16, device="cuda")
model.init_weights(seed=0)
batch = None
prm = dict(model.named_parameters())
EMB = [n for n in prm if n.startswith("img_embeddings.") and "pano_encoder" not in n] + ["embeddings.token_type_embeddings.weight"]
ENC = [n for n in prm if "pano_encoder" in n]
REST = [n for n in prm if n not in EMB and n not in ENC and prm[n].numel() <= 100_000]      # biases, LayerNorms, small tables
na = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
nb = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
noise_stream = torch.cuda.Stream()


def dev(snaps, names):
    worst = (0.0, "")
    for n in names:
        stack = torch.stack([r[n].double().reshape(-1) for r in snaps])
        med = stack.median(0).values
        scale = max(float(med.abs().max()), 1e-6)
        worst = max(worst, (float((stack - med).abs().max()) / scale, n))
    return worst


def bytes_diff(bufs, plans=0):
    """-> (number of differing bytes against repetition 0 over all repetitions, first / last differing byte offset)"""
    ref = bufs[0]
    diff = torch.zeros_like(ref, dtype=torch.bool)
    for b_ in bufs[1:]:
        diff |= b_ != ref
    nz = diff.nonzero().reshape(-1)
    out = [int(nz.numel()), int(nz.min()) if nz.numel() else -1, int(nz.max()) if nz.numel() else -1, ref.numel()]
    if plans and nz.numel():         # the workspace is `plans` equal bump plans (planner.hip plan_pano_ws): final norm, layer 1, layer 0, embed
        size = (ref.numel() - 256) // plans
        for k in range(plans):
            m = nz[(nz >= k * size) & (nz < (k + 1) * size)] - k * size
            if m.numel():
                out.append(f"plan {k}: {int(m.numel())} bytes in [{int(m.min())}, {int(m.max())}] of {size}")
    return tuple(out)


def run_mode(tag, overlap, lds, drain, noise):
    os.environ["ETP_PANO_BWD_LDS"] = str(lds)
    os.environ["ETP_PANO_BWD_DRAIN"] = str(drain)
    step = PlannerStep(model, batch, overlap=overlap, dropout=None, drop_seed=9)
    snaps, wss, sts = [], [], []
    for _ in range(RUNS):
        step.step_no = 0
        if noise:
            with torch.cuda.stream(noise_stream):
                for _ in range(60):
                    torch.matmul(na, nb)
        step.run_eager()
        torch.cuda.synchronize()
        snaps.append({n: prm[n].grad.detach().clone() for n in EMB + ENC + REST})
        wss.append(step.ws_pano.view(torch.uint8).clone())
        sts.append(step.st_pano.view(torch.uint8).clone())
    e, c, r = dev(snaps, EMB), dev(snaps, ENC), dev(snaps, REST)
    print(f"{tag}: embed-bwd grads {e[0]:.2e} ({e[1]}) | pano-encoder grads {c[0]:.2e} ({c[1]}) | others {r[0]:.2e} ({r[1]})")
    print(f"{tag}:   ws_pano differing bytes / first / last / size {bytes_diff(wss, plans=4)}   st_pano {bytes_diff(sts)}", flush=True)
    step.close()


MODES = [("A three-stream 12K", True, 12288, 0, False), ("D one-stream 12K + foreign GEMMs", False, 12288, 0, True),
         ("B three-stream 12K drain-before", True, 12288, 1, False), ("C three-stream 12K drain-before+after", True, 12288, 2, False),
         ("E one-stream 12K alone", False, 12288, 0, False), ("F three-stream 160K", True, 160 * 1024, 0, False)]
for key in WLS:
    w = dict(bench.WORKLOADS[key])
    assert vars(default_config(w["task"], image_feat_size=w["image_feat_size"])) == vars(cfg), key       # one model serves both
    batch = make_batch(cfg.vocab_size, cfg.image_feat_size, cfg.depth_feat_size, w["B"], w["L"], w["V"], w["G"], seed=1234)
    print(f"== workload {key}: B {w['B']} L {w['L']} V {w['V']} G {w['G']}  (t = {time.time() - t_start:.1f} s)", flush=True)
    for tag, overlap, lds, drain, noise in MODES:
        if time.time() - t_start > float(os.environ.get("BUDGET_S", "120")):
            print(tag, "skipped (time budget)")
            continue
        run_mode(tag, overlap, lds, drain, noise)
print(f"total {time.time() - t_start:.1f} s")

"""Neighbour matrix: every row kernel of the library that keeps per-row state in registers, launched
through the C ABI on one stream WHILE a 128x128-tile GEMM class keeps a second stream busy, 24 repetitions each, compared with the same
launch on a drained device.  Per-row outputs (stores computed from registers: dx, da / dd, dz, dQ / dK / dV, the LayerNorm slabs) must
be BIT-identical; sums that the kernel reduces with atomics (parameter gradients) within 2e-5 of their abs-max (their summation order
is not fixed even alone).

Round 5 found `pano_embed_bwd` (and round 4 `gmap_embed_bwd`) returning different results beside the 128x128 tile class of either GEMM
family -- 24 of 24 repetitions in a standalone reproducer (tools/experiments/r05_pano_bwd_neighbours.cpp) -- and shipped a
CU-exclusive launch for it (a workgroup that asks for the CU's whole LDS shares it with nobody).  Round 6 ran this matrix, bisected
the aggressor and the victim (profiles/r06_neighbour_bisect.txt) and isolated ONE instruction form in a synthetic reproducer
(profiles/r06_pk_opsel_repro.txt): `v_pk_mul_f32 ... op_sel:[0,1]` returns 0 in its low half for lanes 48-63 while another kernel's
wavefronts issue MFMAs on the CU.  The row-kernel objects are now compiled without packed fp32 (build-time audit), every family runs
SHARED by default, and this file keeps it that way:

  * test_default_launches_are_clean_beside_the_128x128_classes  ASSERTS: with the library's default mask (launch.h ROWF_DEFAULT = 0:
    every family shares its CUs) no victim deviates beside any aggressor.  That is the product's guarantee.
  * test_exclusive_launch_switch_still_works: the CU-exclusive launch (mask 255) as a switch.

Reference sites of the victims: BertLayerNorm (vilmodel_cmt.py:150-154,189-193), BertEmbeddings (:62-77), ImageEmbeddings (:695-711),
gmap embedding (:728-730), NextActionPrediction (:651-661), BertSelfAttention / BertOutAttention (:103-141,325-352)."""
import ctypes
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from etpnav_amd import _lib  # noqa: E402
from etpnav_amd._lib import AttnBwdDesc, AttnDesc, GemmDesc, check, ptr  # noqa: E402

DEV = "cuda"
BF, F32 = _lib.ETP_BF16, _lib.ETP_F32
T = torch.bfloat16
H = 768
REPS = 24
# family bits of csrc/launch.h RowFamily
ROWF = dict(pano_embed_bwd=1, gmap_embed_bwd=2, text_embed_bwd=4, sap_tail_bwd=8, ln_bwd_s=16, ln_fwd_s=32, attn_rows_bwd=64,
            attn_rows_fwd=128)
ROWF_DEFAULT = 0           # keep in step with launch.h (checked by test_default_mask_matches_the_header)


def L():
    return _lib.lib()


class Victim:
    """`run(stream)` enqueues the launch(es); `exact` tensors are compared bitwise, `sums` within 2e-5 of abs-max."""

    def __init__(self, name, family, run, exact, sums=(), zero=()):
        self.name, self.family, self.run, self.exact, self.sums, self.zero = name, family, run, list(exact), list(sums), list(zero)

    def launch(self, st):
        with torch.cuda.stream(st):
            for t in self.zero:
                t.zero_()
            for t in self.exact:
                t.fill_(float("nan")) if t.dtype.is_floating_point else t.zero_()
            self.run(st.cuda_stream)

    def snapshot(self):
        return [t.clone() for t in self.exact], [t.clone() for t in self.sums]


def make_victims():
    torch.manual_seed(6)
    v = []
    M = 2560
    # ---- LayerNorm forward / backward on the fp32 stream (text rows of config 2)
    x = torch.randn(M, H, device=DEV) * 2 + 0.3
    g = torch.rand(H, device=DEV) + 0.5
    b = torch.randn(H, device=DEV)
    y = torch.empty(M, H, device=DEV); ylp = torch.empty(M, H, device=DEV, dtype=T); stats = torch.empty(M, 2, device=DEV)
    v.append(Victim("ln_fwd_s", "ln_fwd_s",
                    lambda s: check(L().etp_ln_stream_fwd(BF, ptr(x), ptr(g), ptr(b), ptr(y), ptr(ylp), ptr(stats), M, H, 1e-12, s), "ln_fwd"),
                    exact=[y, ylp, stats]))
    stats0 = torch.stack([x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-12)], 1).contiguous()
    dy = torch.randn(M, H, device=DEV); add = torch.randn(M, H, device=DEV)
    dx = torch.empty(M, H, device=DEV); dxl = torch.empty(M, H, device=DEV, dtype=T)
    dg = torch.zeros(H, device=DEV); db = torch.zeros(H, device=DEV)
    part = torch.empty(int(L().etp_ln_bwd_part_bytes(M, H)) // 4, device=DEV)
    v.append(Victim("ln_bwd_s", "ln_bwd_s",
                    lambda s: check(L().etp_ln_stream_bwd_stage1(BF, ptr(dy), ptr(x), ptr(stats0), ptr(g), ptr(add), ptr(dx), ptr(dxl),
                                                                 ptr(dg), ptr(db), ptr(part), M, H, s), "ln_bwd_s"),
                    exact=[dx, dxl, part]))
    # ---- text embedding backward (B = 32, L = 80)
    B, Lt, vocab = 32, 80, 3000
    word = torch.randn(vocab, H, device=DEV); pos = torch.randn(Lt + 3, H, device=DEV); typ = torch.randn(2, H, device=DEV)
    ids = torch.randint(1, vocab, (B, Lt), device=DEV)
    te_y = torch.empty(B * Lt, H, device=DEV); te_st = torch.empty(B * Lt, 2, device=DEV)
    check(L().etp_text_embed_fwd(F32, ptr(ids), ptr(word), ptr(pos), ptr(typ), ptr(g), ptr(b), ptr(te_y), None, ptr(te_st), B, Lt, H,
                                 1e-12, torch.cuda.current_stream().cuda_stream), "text_embed_fwd")
    te_dy = torch.randn(B * Lt, H, device=DEV) * 0.01
    dword = torch.zeros_like(word); dpos = torch.zeros_like(pos); dtyp = torch.zeros(H, device=DEV)
    tdg = torch.zeros(H, device=DEV); tdb = torch.zeros(H, device=DEV)
    v.append(Victim("text_embed_bwd", "text_embed_bwd",
                    lambda s: check(L().etp_text_embed_bwd(F32, ptr(te_dy), ptr(ids), ptr(word), ptr(pos), ptr(typ), ptr(g), ptr(te_st),
                                                           ptr(dword), ptr(dpos), ptr(dtyp), ptr(tdg), ptr(tdb), B, Lt, H, s), "text_embed_bwd"),
                    exact=[], sums=[dword, dpos, dtyp, tdg, tdb], zero=[dword, dpos, dtyp, tdg, tdb]))
    # ---- graph-node embedding backward (B * G = 512 rows)
    Mg = 512
    img = torch.randn(Mg, H, device=DEV); step_ids = torch.randint(0, 20, (Mg,), device=DEV); posf = torch.randn(Mg, 7, device=DEV)
    step_emb = torch.randn(100, H, device=DEV); w_pos = torch.randn(H, 7, device=DEV) * 0.3; b_pos = torch.randn(H, device=DEV) * 0.02
    gx = torch.empty(Mg, H, device=DEV); gxl = torch.empty(Mg, H, device=DEV, dtype=T); gst = torch.empty(Mg, 2, device=DEV)
    check(L().etp_gmap_embed_fwd(BF, ptr(img), ptr(step_ids), ptr(posf), ptr(step_emb), ptr(w_pos), ptr(b_pos), ptr(g), ptr(b), ptr(gx),
                                 ptr(gxl), ptr(gst), Mg, H, 7, torch.cuda.current_stream().cuda_stream), "gmap_embed_fwd")
    gdx = torch.randn(Mg, H, device=DEV) * 0.01
    d_step = torch.zeros_like(step_emb); d_w = torch.zeros_like(w_pos); d_b = torch.zeros(H, device=DEV)
    gdg = torch.zeros(H, device=DEV); gdb = torch.zeros(H, device=DEV)
    v.append(Victim("gmap_embed_bwd", "gmap_embed_bwd",
                    lambda s: check(L().etp_gmap_embed_bwd(BF, ptr(gdx), ptr(step_ids), ptr(posf), ptr(w_pos), ptr(b_pos), ptr(g), ptr(gst),
                                                           ptr(d_step), ptr(d_w), ptr(d_b), ptr(gdg), ptr(gdb), Mg, H, 7, s), "gmap_embed_bwd"),
                    exact=[], sums=[d_step, d_w, d_b, gdg, gdb], zero=[d_step, d_w, d_b, gdg, gdb]))
    # ---- SAP head tail backward (512 node rows)
    r = torch.relu(torch.randn(Mg, H, device=DEV)).to(T)
    w2 = torch.randn(H, device=DEV) * 0.05; b2 = torch.zeros(1, device=DEV)
    visited = (torch.rand(Mg, device=DEV) < 0.2).to(torch.uint8); valid = (torch.rand(Mg, device=DEV) < 0.9).to(torch.uint8)
    logits = torch.empty(Mg, device=DEV); sst = torch.empty(Mg, 2, device=DEV)
    check(L().etp_sap_tail_fwd(BF, ptr(r), ptr(g), ptr(b), ptr(w2), ptr(b2), ptr(visited), ptr(valid), ptr(logits), ptr(sst), Mg, H,
                               torch.cuda.current_stream().cuda_stream), "sap_tail_fwd")
    dlog = torch.randn(Mg, device=DEV) * 0.1
    dz = torch.empty(Mg, H, device=DEV, dtype=T)
    sdg = torch.zeros(H, device=DEV); sdb = torch.zeros(H, device=DEV); dw2 = torch.zeros(H, device=DEV); db2 = torch.zeros(1, device=DEV)
    v.append(Victim("sap_tail_bwd", "sap_tail_bwd",
                    lambda s: check(L().etp_sap_tail_bwd(BF, ptr(dlog), ptr(r), ptr(g), ptr(b), ptr(w2), ptr(sst), ptr(visited), ptr(valid),
                                                         ptr(dz), ptr(sdg), ptr(sdb), ptr(dw2), ptr(db2), Mg, H, s), "sap_tail_bwd"),
                    exact=[dz], sums=[sdg, sdb, dw2, db2], zero=[sdg, sdb, dw2, db2]))
    # ---- panorama view-embedding backward (B * V = 1152 rows): the three launches of round 5
    Mp = 1152
    a = torch.randn(Mp, H, device=DEV).to(T); d = torch.randn(Mp, H, device=DEV).to(T)
    loc = torch.randn(Mp, 4, device=DEV); nav = (torch.arange(Mp, device=DEV) % 36 < 4).long()
    psz = [H, H, H, H, 4 * H, H, H, H, 2 * H, H, H, H]
    params = [(1.0 + 0.1 * torch.randn(n, device=DEV)) if i in (0, 2, 6, 10) else (0.3 if i == 4 else 0.02) * torch.randn(n, device=DEV)
              for i, n in enumerate(psz)]
    grads = [torch.zeros(n, device=DEV) for n in psz]
    PP = (ctypes.c_void_p * 12)(*[p.data_ptr() for p in params]); GG = (ctypes.c_void_p * 12)(*[p.data_ptr() for p in grads])
    py = torch.empty(Mp, H, device=DEV); pst = torch.empty(Mp, 8, device=DEV)
    check(L().etp_pano_embed_fwd(BF, ptr(a), ptr(d), ptr(loc), ptr(nav), PP, ptr(py), ptr(pst), Mp, H,
                                 torch.cuda.current_stream().cuda_stream), "pano_embed_fwd")
    pdy = torch.randn(Mp, H, device=DEV) * 0.01
    da = torch.empty(Mp, H, device=DEV, dtype=T); dd = torch.empty(Mp, H, device=DEV, dtype=T)
    v.append(Victim("pano_embed_bwd", "pano_embed_bwd",
                    lambda s: check(L().etp_pano_embed_bwd(BF, ptr(pdy), ptr(a), ptr(d), ptr(loc), ptr(nav), ptr(pst), PP, GG, ptr(da), ptr(dd),
                                                           Mp, H, s), "pano_embed_bwd"),
                    exact=[da, dd], sums=grads, zero=grads))
    v[-1]._keep = (params, PP, GG)
    # ---- register-resident attention backward: text self-attention (80 x 80: rows_bwd<5>) and graph self-attention with the
    # pairwise-distance bias (16 x 16: rows_bwd<1, true>)
    for name, Bq, nh, Lq, Lk, with_dist in (("attn_rows_bwd<5> 80x80", 32, 12, 80, 80, False), ("attn_rows_bwd<1,dist> 16x16", 32, 12, 16, 16, True)):
        Hh = nh * 64
        ldS = (Lk + 7) // 8 * 8
        q = torch.randn(Bq * Lq, Hh, device=DEV).to(T); kv = torch.randn(Bq * Lk, 2 * Hh, device=DEV).to(T)
        km = torch.rand(Bq, Lk, device=DEV) > 0.2
        km[:, 0] = True
        dist = torch.rand(Bq, Lq, Lk, device=DEV); w = torch.tensor([0.3], device=DEV); b0 = torch.tensor([0.1], device=DEV)
        P = torch.empty(Bq, nh, Lq, ldS, device=DEV, dtype=T); ctx = torch.empty(Bq * Lq, Hh, device=DEV, dtype=T)
        fd = AttnDesc()
        fd.dtype, fd.B, fd.heads, fd.Lq, fd.Lk, fd.ldS = BF, Bq, nh, Lq, Lk, ldS
        fd.Q, fd.ldq = q.data_ptr(), Hh
        fd.K, fd.ldk = kv.data_ptr(), 2 * Hh
        fd.V, fd.ldv = kv.data_ptr() + Hh * 2, 2 * Hh
        fd.P, fd.ctx, fd.ldc = P.data_ptr(), ctx.data_ptr(), Hh
        fd.keymask, fd.mask_mode = km.data_ptr(), 0
        if with_dist:
            fd.dist, fd.sp_w, fd.sp_b = dist.data_ptr(), w.data_ptr(), b0.data_ptr()
        fd.alpha = 0.125
        check(L().etp_attn_fwd(ctypes.byref(fd), torch.cuda.current_stream().cuda_stream), "attn_fwd")
        dctx = torch.randn(Bq * Lq, Hh, device=DEV).to(T)
        dP = torch.empty_like(P); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
        dw = torch.zeros(1, device=DEV); dbb = torch.zeros(1, device=DEV)
        bd = AttnBwdDesc()
        bd.f = fd
        bd.dctx, bd.ldd, bd.dP = dctx.data_ptr(), Hh, dP.data_ptr()
        bd.dQ, bd.lddq = dq.data_ptr(), Hh
        bd.dK, bd.lddk = dkv.data_ptr(), 2 * Hh
        bd.dV, bd.lddv = dkv.data_ptr() + Hh * 2, 2 * Hh
        if with_dist:
            bd.d_sp_w, bd.d_sp_b = dw.data_ptr(), dbb.data_ptr()
        vv = Victim(name, "attn_rows_bwd", (lambda bd_: (lambda s: check(L().etp_attn_bwd(ctypes.byref(bd_), s), "attn_bwd")))(bd),
                    exact=[dq, dkv], sums=[dw, dbb] if with_dist else [], zero=[dw, dbb] if with_dist else [])
        vv._keep = (q, kv, km, dist, w, b0, P, ctx, dctx, dP, fd, bd)
        v.append(vv)
    torch.cuda.synchronize()
    return v


def make_aggressors():
    """name -> (callable(stream_handle) enqueuing ONE launch, library switches to set while it runs)"""
    torch.manual_seed(7)
    GM, GN, GK = 2560, 3072, 768
    A = torch.randn(GM, GK, device=DEV).to(T); Bw = (torch.randn(GN, GK, device=DEV) * 0.05).to(T)
    C = torch.empty(GM, GN, device=DEV, dtype=T)

    def desc(a, b, c, M, N, K, ta, tb, cdt):
        d = GemmDesc()
        d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
        d.M, d.N, d.K = M, N, K
        d.lda, d.ldb, d.ldc = a.stride(0), b.stride(0), c.stride(0)
        d.trans_a, d.trans_b, d.dtype, d.c_dtype = ta, tb, BF, cdt
        d.batch, d.batch_inner, d.ksplit, d.alpha = 1, 1, 1, 1.0
        return d

    d_nt = desc(A, Bw, C, GM, GN, GK, 0, 0, BF)
    # grouped weight gradients of one text layer (TN, fp32 out): dW[N, K] = dY[M, N]^T X[M, K]
    dY = torch.randn(GM, 3072, device=DEV).to(T); X = torch.randn(GM, 768, device=DEV).to(T)
    dWs = [torch.empty(n, 768, device=DEV) for n in (2304, 768, 3072)]
    grp = (GemmDesc * 3)(*[desc(dY[:, :n], X, w_, n, 768, GM, 1, 1, F32) for n, w_ in zip((2304, 768, 3072), dWs)])
    keep = (A, Bw, C, dY, X, dWs, grp, d_nt)
    return {
        "mm32 128x128 (bf16 2560x3072x768 NT)": (lambda s: check(L().etp_gemm(ctypes.byref(d_nt), s), "gemm"), {}, keep),
        "gemm.hip 128x128 (same product, MM32=0)": (lambda s: check(L().etp_gemm(ctypes.byref(d_nt), s), "gemm"), {"MM32": "0"}, keep),
        "mm32 grouped 128x128 (text-layer weight gradients, TN fp32)": (lambda s: check(L().etp_gemm_group(grp, 3, s), "gemm_group"), {}, keep),
    }


def run_matrix(victims, aggressors, mask):
    """-> {victim: {aggressor: {"bad_reps": n, "reps": REPS, "worst_sum": x, "rows": max differing rows}}}"""
    _lib.set_option("ROW_EXCLUSIVE", str(mask))
    s_gemm = torch.cuda.Stream(priority=-1)
    s_vic = torch.cuda.Stream(priority=0)
    out = {}
    for v in victims:
        torch.cuda.synchronize()
        v.launch(s_vic)
        torch.cuda.synchronize()
        ref_exact, ref_sums = v.snapshot()
        # alone, again: the drained-device launch must reproduce itself (otherwise the comparison below means nothing)
        v.launch(s_vic)
        torch.cuda.synchronize()
        for t, r0 in zip(v.exact, ref_exact):
            assert torch.equal(t.view(torch.uint8), r0.view(torch.uint8)), (v.name, "not bit-reproducible ALONE")
        res = {}
        for aname, (agg, switches, _keep) in aggressors.items():
            for k, val in switches.items():
                _lib.set_option(k, val)
            bad, worst, rows_max = 0, 0.0, 0
            try:
                for _ in range(REPS):
                    torch.cuda.synchronize()
                    for j in range(24):
                        agg(s_gemm.cuda_stream)
                        if j == 2:
                            v.launch(s_vic)          # goes out while the neighbour stream is busy and stays busy
                    torch.cuda.synchronize()
                    dirty = False
                    for t, r0 in zip(v.exact, ref_exact):
                        if not torch.equal(t.view(torch.uint8), r0.view(torch.uint8)):
                            dirty = True
                            if t.dim() == 2:
                                rows_max = max(rows_max, int((t.view(torch.uint8) != r0.view(torch.uint8)).any(1).sum()))
                    for t, r0 in zip(v.sums, ref_sums):
                        e = float((t - r0).abs().max()) / max(float(r0.abs().max()), 1e-20)
                        worst = max(worst, e)
                        if e > 2e-5:
                            dirty = True
                    bad += dirty
            finally:
                for k in switches:
                    _lib.set_option(k, None)
            res[aname] = {"bad_reps": bad, "reps": REPS, "worst_sum": worst, "rows": rows_max}
        out[v.name] = res
    _lib.set_option("ROW_EXCLUSIVE", None)
    return out


def _dump(name, obj):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        json.dump(obj, open(os.path.join(d, name), "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def test_default_mask_matches_the_header():
    src = open(os.path.join(os.path.dirname(_lib.HEADER), "..", "etpnav_amd", "csrc", "launch.h")).read()
    import re
    bits = {k: v for k, v in re.findall(r"ROWF_(\w+) = (\d+)", src) if k != "DEFAULT"}
    assert {k.lower(): int(v) for k, v in bits.items()} == {"pano_bwd": 1, "gmap_bwd": 2, "text_bwd": 4, "sap_bwd": 8, "ln_bwd": 16,
                                                            "ln_fwd": 32, "attn_bwd": 64, "attn_fwd": 128}
    dflt = re.search(r"^\s*ROWF_DEFAULT = ([^\n}]+)", src, re.M).group(1).strip()       # the enum line, not the comment above it
    val = 0
    for term in dflt.split("|"):
        term = term.strip()
        val |= int(term) if term.isdigit() else int(bits[term[5:]])
    assert val == ROWF_DEFAULT, (dflt, ROWF_DEFAULT)


@pytest.fixture(scope="module")
def setup():
    return make_victims(), make_aggressors()


def test_default_launches_are_clean_beside_the_128x128_classes(setup):
    victims, aggressors = setup
    m = run_matrix(victims, aggressors, ROWF_DEFAULT)
    _dump("r06_neighbour_matrix_default.json", m)
    dirty = {(v, a): r for v, row in m.items() for a, r in row.items() if r["bad_reps"]}
    assert not dirty, dirty


def test_exclusive_launch_switch_still_works(setup):
    """ROW_EXCLUSIVE = 255: every family asks for the CU's whole LDS (launch.h row_launch_lds: size from the device's properties, the
    attribute set per device) -- the launches must go through and give the same results, beside the first aggressor."""
    victims, aggressors = setup
    first = dict(list(aggressors.items())[:1])
    m = run_matrix(victims, first, 255)
    _dump("r06_neighbour_matrix_exclusive.json", m)
    dirty = {(v, a): r for v, row in m.items() for a, r in row.items() if r["bad_reps"]}
    assert not dirty, dirty

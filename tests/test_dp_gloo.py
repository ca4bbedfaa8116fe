"""Data-parallel gradient averaging (etpnav_amd/dp.py) with world_size 2 on CPU (gloo backend): dense buckets in
fp32 and bf16 transport, plus the row-sparse word-embedding exchange, against the analytically known mean."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from etpnav_amd import dp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _ids_of(r, ragged):
    # ragged: the ranks touch DIFFERENT numbers of rows (the reference's collate pads to the per-batch maximum, so B*L differs
    # across ranks: ADVICE r2) -- the exchange must use a rank-independent capacity, not len(row_ids)
    return torch.tensor([3 + r, 7, 7, 20 + 2 * r] + ([31, 7, 40] if (ragged and r == 1) else []))


def _worker(rank, world, port, comm_dtype, q, ragged=False, capacity=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        n_rows, row_len = 50, 8
        n = 1000 + n_rows * row_len
        grads = [torch.randn(n) for _ in range(world)]          # every rank can rebuild all ranks' gradients
        for r in range(world):
            g = torch.Generator().manual_seed(100 + r)
            grads[r] = torch.randn(n, generator=g)
            table = grads[r][1000:].view(n_rows, row_len)
            ids_r = _ids_of(r, ragged)
            mask = torch.zeros(n_rows, dtype=torch.bool); mask[ids_r] = True
            table[~mask] = 0                                     # row-sparse, as an embedding gradient is
        mine = grads[rank].clone()
        ids = _ids_of(rank, ragged)
        red = dp.GradReducer(mine, [(600, 1000), (0, 600)], comm_dtype=comm_dtype, sparse_rows=(1000, n_rows, row_len))
        red.reduce_bucket(0)
        red.reduce_bucket(1)
        red.reduce_sparse_rows(ids, capacity=capacity)
        red.finish()
        if comm_dtype == torch.float32:
            expect = sum(grads) / world
            tol = 1e-6
        else:
            expect = sum(g.to(comm_dtype).float() for g in grads) / world
            tol = 2e-2
        err = (mine - expect).abs().max().item()
        q.put((rank, err <= tol, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("comm_dtype,ragged,capacity", [(torch.float32, False, None), (torch.bfloat16, False, None),
                                                        (torch.float32, True, 12), (torch.float32, True, None)])
def test_grad_reducer_world2_gloo(comm_dtype, ragged, capacity):
    """ragged + capacity=12: fixed rank-independent block size; ragged + capacity=None: the maximum length is agreed with
    one scalar all-reduce.  Both must give the dense mean although rank 1 touches 7 rows and rank 0 four."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, comm_dtype, q, ragged, capacity)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert ok, f"rank {rank}: max err {err}"


@pytest.mark.parametrize("world,comm_dtype,ragged,capacity", [(4, torch.float32, True, 12), (4, torch.bfloat16, False, None),
                                                              (8, torch.float32, True, None), (8, torch.float32, False, 16)])
def test_grad_reducer_world4_and_world8_gloo(world, comm_dtype, ragged, capacity):
    """The same reducer over 4 and 8 ranks (the driver's SCALE line runs 1 / 2 / 4 / 8; no box with more than one GPU has been
    available to the builder in five rounds): dense buckets + the row-sparse table exchange with rank-dependent row counts, a
    shared id (7) on every rank, a repeated id per rank, fixed and agreed capacities."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, comm_dtype, q, ragged, capacity)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert len(res) == world
    for rank, ok, err in res:
        assert ok, f"world {world} rank {rank}: max err {err}"


def test_native_bucket_slice_arithmetic_for_worlds_2_4_8():
    """etp_allreduce_plan (csrc/comm.hip): the per-rank slice / tail / bf16 pad arithmetic etp_allreduce_bucket_ready applies, as a pure
    host function through the C ABI, on the bucket sizes the planner really produces (dp.planner_buckets_layered of the BERT-base and
    XLM-R planners, 9 text groups as bench.py uses, with and without frozen sub-models) and on adversarial sizes -- then the
    collective sequence is replayed on the host for W simulated ranks and must give the mean:
      fp32: reduce-scatter(sum) of [0, body) into slice r -> scale slice by 1/W -> all-gather; all-reduce(sum) of the tail, scale
      bf16: stage as bf16 with a zeroed pad -> reduce-scatter + all-gather over per * W -> unpack * 1/W."""
    import ctypes
    from etpnav_amd import _lib
    from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
    L = _lib.lib()
    sizes = {1, 63, 64, 65, 511, 512, 513, 4095, 4096, 8 * 64 * 3 + 17, 1 << 20}
    for kw in (dict(), dict(fix_lang_embedding=True), dict(task_type="rxr")):
        task = kw.pop("task_type", "r2r")
        m = GlocalTextPathNavCMT(default_config(task, **kw), dtype=torch.float32, device="cpu")
        ranges, _, _ = dp.planner_buckets_layered(m, 9)
        sizes |= {e - s for s, e in ranges}
        del m
    out = (ctypes.c_int64 * 4)()
    for W in (2, 4, 8):
        for n in sorted(sizes):
            for dt in (_lib.ETP_F32, _lib.ETP_BF16):
                assert L.etp_allreduce_plan(n, W, dt, out) == 0
                per, body, tail, staged = (int(x) for x in out)
                if dt == _lib.ETP_F32:
                    assert per % 64 == 0 and body == per * W and body + tail == n and 0 <= tail < W * 64 and staged == 0, (W, n, list(out))
                else:
                    assert per % 8 == 0 and body == per * W == staged and tail == 0 and n <= staged < n + W * 8, (W, n, list(out))
                    assert staged <= int(L.etp_allreduce_staging_elems(n, W)), (W, n)
        # replay on the host (small buckets only: the arithmetic is size-independent)
        for n in (1, 65, 513, 8 * 64 * 3 + 17):
            g = torch.Generator().manual_seed(n + W)
            ranks = [torch.randn(n, generator=g) for _ in range(W)]
            mean = sum(ranks) / W
            L.etp_allreduce_plan(n, W, _lib.ETP_F32, out)
            per, body, tail, _ = (int(x) for x in out)
            res = [r.clone() for r in ranks]
            for r in range(W):                                   # reduce-scatter + local scale
                res[r][r * per:(r + 1) * per] = sum(x[r * per:(r + 1) * per] for x in ranks) / W
            for r in range(W):                                   # all-gather
                for q_ in range(W):
                    res[r][q_ * per:(q_ + 1) * per] = res[q_][q_ * per:(q_ + 1) * per]
                res[r][body:] = sum(x[body:] for x in ranks) / W     # tail all-reduce
                assert torch.allclose(res[r], mean, atol=1e-6), (W, n, r)
            L.etp_allreduce_plan(n, W, _lib.ETP_BF16, out)
            per, body, _, staged = (int(x) for x in out)
            st = [torch.cat([r.to(torch.bfloat16), torch.zeros(staged - n, dtype=torch.bfloat16)]) for r in ranks]
            red = sum(x.float() for x in st).to(torch.bfloat16)      # what the gathered copy holds on every rank
            assert red.numel() == per * W
            got = red[:n].float() / W
            assert (got - mean).abs().max().item() <= 3e-2 * max(1.0, mean.abs().max().item()), (W, n)


def _native_decision_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            comm = dp.NativeComm.create(torch.device("cpu"), torch.float32, 1024)
        q.put((rank, comm is None))
    finally:
        dist.destroy_process_group()


def test_native_communicator_decision_is_collective():
    """a rank that cannot set up the library's RCCL communicator must not leave the others inside
    ncclCommInitRank.  On CPU ranks the initialisation cannot succeed; NativeComm.create must come back None on EVERY rank
    (availability and outcome are min-reduced over the group) without hanging, and the reducer then uses torch.distributed."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_decision_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(none for _, none in res), res


def _selftest_decision_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        from etpnav_amd import _lib
        log = []

        class FakeComm(dp.NativeComm):                       # initialises fine everywhere; the self-test fails on rank 1 only
            def __init__(self, device, comm_dtype=torch.float32, max_bucket_elems=0, group=None):
                self.rank, self.world, self.handle = dist.get_rank(group), dist.get_world_size(group), None

            def self_test(self, timeout_s=60.0):
                log.append(("self_test", timeout_s))
                return "collectives did not complete within 1 s" if self.rank == 1 else None

            def abort(self):
                log.append(("abort",))

            def close(self):
                log.append(("close",))

        class FakeLib:
            @staticmethod
            def etp_allreduce_available():
                return 1

        real = _lib.lib
        _lib.lib = lambda: FakeLib
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                comm = FakeComm.create(torch.device("cpu"), torch.float32, 1024)
        finally:
            _lib.lib = real
        q.put((rank, comm is None, [e[0] for e in log], any("self-test failed" in str(x.message) for x in w)))
    finally:
        dist.destroy_process_group()


def test_failed_first_contact_self_test_disables_the_native_path_on_every_rank():
    """The known-answer / deadline self-test of a freshly initialised communicator (NativeComm.self_test) fails on ONE rank:
    every rank must abort its communicator and come back with None (-> torch.distributed collectives), the healthy rank too."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_selftest_decision_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, none, calls, warned in res:
        assert none and warned, res
        assert calls[:2] == ["self_test", "abort"], res


def test_planner_bucket_ranges_cover_the_arena_once():
    from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
    m = GlocalTextPathNavCMT(default_config(vocab_size=1024), dtype=torch.float32, device="cpu")
    ranges, (woff, wrows, wlen) = dp.planner_buckets(m)
    cover = torch.zeros(m.flat_grads.numel(), dtype=torch.int32)
    for s, e in ranges:
        cover[s:e] += 1
    cover[woff:woff + wrows * wlen] += 1
    for name, p in m.named_parameters():
        off = (p.data_ptr() - m.flat_params.data_ptr()) // 4
        assert bool((cover[off:off + p.numel()] == 1).all()), name
    # bucket 0 (reduced while the text backward runs) must not contain text-encoder gradients
    s0, e0 = ranges[0]
    for name, p in m.named_parameters():
        off = (p.data_ptr() - m.flat_params.data_ptr()) // 4
        if name.startswith("lang_encoder.") or name.startswith("embeddings."):
            assert not (s0 <= off < e0), name


def test_layered_buckets_cover_the_arena_once_and_follow_backward_order():
    from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
    m = GlocalTextPathNavCMT(default_config(vocab_size=1024), dtype=torch.float32, device="cpu")
    ranges, (woff, wrows, wlen), groups = dp.planner_buckets_layered(m, text_groups=3)
    assert groups == [(6, 9), (3, 6), (0, 3)]
    cover = torch.zeros(m.flat_grads.numel(), dtype=torch.int32)
    for s, e in ranges:
        cover[s:e] += 1
    cover[woff:woff + wrows * wlen] += 1
    base = m.flat_params.data_ptr()
    for name, p in m.named_parameters():
        off = (p.data_ptr() - base) // 4
        assert bool((cover[off:off + p.numel()] == 1).all()), name
    # text bucket k holds exactly the matrices of its layer group
    for k, (lo, hi) in enumerate(groups):
        s, e = ranges[1 + k]
        for name, p in m.named_parameters():
            if name.startswith("lang_encoder.layer.") and p.dim() == 2:
                layer = int(name.split(".")[2])
                off = (p.data_ptr() - base) // 4
                assert (s <= off < e) == (lo <= layer < hi), name


def test_pretraining_variant_buckets_cover_language_side_weights_and_dense_word_table():
    """Pre-training model (use_lang2visn_attn): the language-side x-layer weights and the MLM head land in the non-text
    matrix / vector buckets, and with dense_word_table (MLM: tied decoder) the word table is one more dense range."""
    from etpnav_amd.planner import GlocalTextPathNavCMT, default_config
    m = GlocalTextPathNavCMT(default_config(vocab_size=1024, use_lang2visn_attn=True, num_l_layers=2, num_x_layers=2),
                             dtype=torch.float32, device="cpu")
    ranges, sparse = dp.planner_buckets(m, dense_word_table=True)
    assert sparse is None
    cover = torch.zeros(m.flat_grads.numel(), dtype=torch.int32)
    for s, e in ranges:
        cover[s:e] += 1
    seen_lang = seen_head = False
    for name, p in m.named_parameters():
        off = (p.data_ptr() - m.flat_params.data_ptr()) // 4
        assert bool((cover[off:off + p.numel()] == 1).all()), name
        seen_lang |= ".lang_self_att." in name
        seen_head |= name.startswith("mlm_head.")
    assert seen_lang and seen_head


def test_per_task_bucket_sets_match_the_parameters_each_pretraining_task_touches():
    """static per-task bucket sets instead of DDP's find_unused_parameters=True (utils/misc.py:58).  The
    oracle (pinned to the real GlocalTextPathCMTPreTraining) says which parameters get a gradient in each task: exactly
    those must lie inside task_grad_ranges(task) (or the sparse word table), everything outside must be zero."""
    from oracle import planner_oracle as po
    from oracle.make_golden_pretrain import make_case
    from etpnav_amd.planner import GlocalTextPathNavCMT
    cfg, P, batch = make_case()
    m = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device="cpu")
    base = m.flat_params.data_ptr()
    offs = {n: ((p.data_ptr() - base) // 4, p.numel()) for n, p in m.named_parameters()}
    for task, fn in (("sap", po.sap_step_with_grads), ("mlm", po.mlm_step_with_grads)):
        _, grads = fn(P, cfg, batch)
        ranges, sparse = dp.task_grad_ranges(m, task)
        assert (sparse is None) == (task == "mlm")
        cover = torch.zeros(m.flat_grads.numel(), dtype=torch.bool)
        for s, e in ranges:
            assert 0 <= s < e <= cover.numel()
            cover[s:e] = True
        if sparse is not None:
            cover[sparse[0]:sparse[0] + sparse[1] * sparse[2]] = True
        n_used = n_unused = 0
        for name, g in grads.items():
            if name.startswith("__input__"):
                continue
            off, n = offs[name]
            touched = bool(g.abs().max() > 0)
            inside = bool(cover[off:off + n].all())
            if touched:
                assert inside, f"{task}: {name} has a gradient but is outside the task's buckets"
            if not dp.task_uses_param(task, name):
                assert not touched, f"{task}: {name} is declared unused but the oracle gives it a gradient"
                assert not bool(cover[off:off + n].any()), f"{task}: unused {name} is inside a bucket"
                n_unused += 1
            else:
                n_used += 1
        assert n_used > 100 and n_unused > 10
    with pytest.raises(ValueError):
        dp.task_uses_param("mrc", "x")


def test_meta_loader_mixes_tasks_by_ratio_and_restarts_exhausted_loaders():
    from etpnav_amd.pretrain import MetaLoader
    epochs = []
    loaders = {"mlm": ([{"i": k} for k in range(3)], 1, lambda e: epochs.append(("mlm", e))),
               "sap": ([{"i": 10 + k} for k in range(5)], 3, lambda e: epochs.append(("sap", e)))}
    g = torch.Generator().manual_seed(0)
    ml = MetaLoader(loaders, accum_steps=2, generator=g)
    it = iter(ml)
    seen = [next(it) for _ in range(400)]
    names = [t for t, _ in seen]
    for k in range(0, 400, 2):
        assert names[k] == names[k + 1]                      # the task only changes every accum_steps steps
    frac = names.count("sap") / len(names)
    assert 0.65 < frac < 0.85                               # ratio 3 : 1
    sap_items = [b["i"] for t, b in seen if t == "sap"]
    assert sap_items[:7] == [10, 11, 12, 13, 14, 10, 11]     # exhausted loader re-created, in order
    assert ("sap", 1) in epochs and any(t == "mlm" for t, _ in epochs)
    plain = MetaLoader({"only": [1, 2]})
    it = iter(plain)
    assert [next(it)[1] for _ in range(5)] == [1, 2, 1, 2, 1]

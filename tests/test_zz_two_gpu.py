"""The only tests that give the library's communicator (etp_allreduce_*, csrc/comm.hip) a PEER: two processes on two GPUs over RCCL.

Skipped on one-GPU boxes (every box the builder ever had: the multi-rank path is otherwise covered by the world-1 tests through the real
RCCL binding in tests/test_dp_gpu.py, the first-contact self-test of NativeComm.create and the gloo runs of tests/test_dp_gloo.py).  They
live in the LAST file of the suite on purpose (round 6): the driver runs `pytest -x`, these tests would be RCCL's first contact with more
than one rank, and a failure or a hang there must cost these four cases, not the 370 behind them (until round 5 they came first in
tests/test_dp_gpu.py).  The workers report through a queue with a deadline of 240 / 300 s."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from etpnav_amd import dp  # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _native_worker(rank, world, port, comm_dtype, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        n_rows, row_len = 300, 768
        n_dense = 64 * 4000 + 37
        n = (n_dense + 63) // 64 * 64 + n_rows * row_len
        off = n - n_rows * row_len
        grads, idsets = [], []
        for r in range(world):                                   # every rank can rebuild all ranks' gradients
            g = torch.Generator().manual_seed(100 + r)
            x = torch.randn(n, generator=g)
            ids_r = torch.tensor([3 + r, 7, 7, 20 + 2 * r, 299] + ([31, 7] if r == 1 else []))
            table = x[off:].view(n_rows, row_len)
            mask = torch.zeros(n_rows, dtype=torch.bool); mask[ids_r] = True
            table[~mask] = 0
            grads.append(x); idsets.append(ids_r)
        mine = grads[rank].clone().cuda()
        red = dp.GradReducer(mine, [(1024, n_dense), (0, 1024)], comm_dtype=comm_dtype, sparse_rows=(off, n_rows, row_len))
        native = red.native is not None
        seen = red.native.ranks_seen() if native else 0
        red.reduce_bucket(0)
        red.reduce_bucket(1)
        red.reduce_sparse_rows(idsets[rank].cuda(), capacity=16)
        red.finish()
        torch.cuda.synchronize()
        if comm_dtype == torch.float32:
            expect, tol = sum(grads) / world, 1e-5
        else:
            expect, tol = sum(g.to(torch.bfloat16).float() for g in grads) / world, 5e-2
        expect[n_dense:off] = grads[rank][n_dense:off]           # alignment gap between the buckets: not reduced
        err = (mine.cpu() - expect).abs().max().item()
        red.close()
        q.put((rank, native, seen, err <= tol, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("comm_dtype", [torch.float32, torch.bfloat16])
def test_native_communicator_two_gpus_dense_and_row_sparse_mean(comm_dtype):
    """the multi-rank path of etp_allreduce_* (in-place reduce-scatter -> scale -> all-gather, the all-reduce tail,
    bf16 staging, gather_rows with ragged id counts) on TWO devices, one process per GPU over RCCL, against the analytic mean.
    Skipped on one-GPU boxes; NativeComm.create's first-contact self-test covers the same collectives at start-up there."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, comm_dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, native, seen, ok, err in res:
        assert native and seen == 2, res
        assert ok, f"rank {rank}: max err {err}"



def _native_step_worker(rank, world, port, overlapped, q):
    """The REAL planner step on two GPUs through the library communicator: free-running issue order (side streams handed to the
    reducer) or joined order; the averaged gradient must equal the gradient of one process holding both half-batches."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from oracle import planner_oracle as po      # input/weight generator only (checker side)
        from etpnav_amd.planner import GlocalTextPathNavCMT
        from etpnav_amd.step import PlannerStep
        dev = f"cuda:{rank}"
        cfg = po.PlannerConfig.r2r(vocab_size=4096)
        P = po.init_params(cfg, seed=3)
        B = 8
        full = po.make_batch(cfg, B=B, L=26, V=15, G=9, seed=50, ragged=False)
        full["txt_ids"][0, 3] = full["txt_ids"][B - 1, 5]
        half = {k: v[rank * (B // 2):(rank + 1) * (B // 2)].clone() for k, v in full.items()}
        model = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device=dev)
        model.load_state_dict(P, strict=True)
        model.eval()
        step = PlannerStep(model, half)
        ranges, sparse, groups = dp.planner_buckets_layered(model, text_groups=3)
        red = dp.GradReducer(model.flat_grads, ranges, comm_dtype=torch.float32, sparse_rows=sparse)
        native = red.native is not None
        for _ in range(2):                                       # twice: the second pass runs with warm communicator state
            step.run_data_parallel(groups, lambda i, side: red.reduce_bucket(i, also=side), overlapped=overlapped and red.overlapped)
            for i in range(1 + len(groups), len(red.ranges)):
                red.reduce_bucket(i)
            red.reduce_sparse_rows(step.inp["txt_ids"], capacity=(B // 2) * 30)
            red.finish()
            torch.cuda.synchronize()
        mine = model.flat_grads.clone()
        step.close()
        ok, err = True, 0.0
        if rank == 0:
            model2 = GlocalTextPathNavCMT(cfg.to_dict(), dtype=torch.float32, device=dev)
            model2.load_state_dict(P, strict=True)
            model2.eval()
            st2 = PlannerStep(model2, full)
            st2.run_eager(); torch.cuda.synchronize()
            ref = model2.flat_grads
            err = (mine - ref).abs().max().item()
            ok = err < 2e-5 + 1e-4 * ref.abs().max().item()
            st2.close()
        red.close()
        q.put((rank, native, ok, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("overlapped", [True, False])
def test_native_two_gpu_planner_step_overlapped_and_joined_equal_full_batch_gradient(overlapped):
    """the overlapped bucket ordering of run_data_parallel (communication stream ordered after the side streams) against
    real gradients on two ranks, beside the joined order -- both must give the full-batch gradient."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_step_worker, args=(r, world, port, overlapped, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, native, ok, err in res:
        assert native, res
        assert ok, f"rank {rank}: max err {err}"

"""Data-parallel gradient averaging over RCCL/xGMI (one process per GPU).

Replaces ``DistributedDataParallel(self.policy.net, ...)`` of ss_trainer_ETP.py:208-212 (and pretrain
utils/misc.py:52-65) for the planner: episodes shard across ranks with no activation exchange; the only collective
is the mean of the parameter gradients, once per step.

MI355X-first choices:
  * all gradients already live in ONE flat fp32 arena, so the reduction runs over a few large contiguous buckets
    (no per-parameter bucketing / flattening copies);
  * on GPUs the dense buckets go through the library's own communicator (C ABI ``etp_allreduce_*``, csrc/comm.hip):
    in-place reduce-scatter -> 1/world scaling of the rank's slice -> all-gather on a private stream, RCCL bound at run
    time; without it (CPU / gloo tests, or ``native=False``) the same buckets go through ``torch.distributed``;
  * fp32 transport by default (DDP's numerics); ``comm_dtype=torch.bfloat16`` is opt-in (half the xGMI bytes, bf16 sums);
  * the word-embedding gradient is row-sparse (<= B*L of 30 522 / 250 002 rows are non-zero): instead of all-reducing
    the dense 94 MB / 768 MB table, ranks all-gather a FIXED-capacity (ids, rows) block and scatter-add locally — same
    result as the dense mean up to summation order, and no host synchronisation (duplicates are masked on the device,
    nothing depends on the number of distinct rows); on GPUs it runs on the library's communicator and stream
    (``etp_allreduce_gather_rows``), so exactly one communicator is ever in flight;
  * buckets are issued right after the backward segment that completes them, so the text-encoder backward overlaps the
    reduction of everything computed before it (see ``PlannerStep``/bench.py).
Works unchanged on CPU with the ``gloo`` backend (tests/test_dp_gloo.py, world_size 2).
"""
from __future__ import annotations

import ctypes
import os
import warnings
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class NativeComm:
    """The library's RCCL communicator (include/etpnav_hip.h ``etp_allreduce_*``).  The 128-byte unique id is created on
    rank 0 and distributed through the existing torch.distributed group (any backend).

    Ranks AGREE on using it: a rank that cannot bind librccl must not leave the others blocked inside
    ncclCommInitRank, so availability is min-reduced over the group before anybody initialises, and the outcome of the
    initialisation is min-reduced again; ``NativeComm.create`` returns None on EVERY rank if any rank failed."""

    @staticmethod
    def _all_ok(ok: bool, device, group) -> bool:
        backend = dist.get_backend(group)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))

    @classmethod
    def create(cls, device, comm_dtype=torch.float32, max_bucket_elems: int = 0, group=None):
        from . import _lib
        try:
            avail = bool(_lib.lib().etp_allreduce_available())
        except Exception:
            avail = False
        if not cls._all_ok(avail, device, group):
            warnings.warn("etp_allreduce_*: librccl is not loadable on every rank; all ranks use torch.distributed collectives")
            return None
        comm, err = None, None
        try:
            comm = cls(device, comm_dtype, max_bucket_elems, group)
        except Exception as e:               # noqa: BLE001  (reported below, on every rank)
            err = e
        if not cls._all_ok(comm is not None, device, group):
            if comm is not None:
                comm.close()
            warnings.warn(f"etp_allreduce_init failed on some rank ({err}); all ranks use torch.distributed collectives")
            return None
        # first contact: a small dense bucket and a small row-sparse exchange with known answers, polled from the host with a
        # deadline -- a communicator that returns wrong values or never completes is aborted on EVERY rank (collective
        # decision again) before any gradient goes through it
        why = None
        if os.environ.get("ETP_DP_SELFTEST", "1") != "0":
            try:
                why = comm.self_test(float(os.environ.get("ETP_DP_SELFTEST_TIMEOUT", "60")))
            except Exception as e:           # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
        if not cls._all_ok(why is None, device, group):
            comm.abort()
            warnings.warn(f"etp_allreduce_* self-test failed on some rank ({why or 'another rank'}); all ranks use "
                          "torch.distributed collectives")
            return None
        return comm

    def __init__(self, device: torch.device, comm_dtype=torch.float32, max_bucket_elems: int = 0, group=None):
        from . import _lib
        self._lib, self.L = _lib, _lib.lib()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.comm_dtype = comm_dtype
        ident = (ctypes.c_ubyte * 128)()
        if self.rank == 0:
            _lib.check(self.L.etp_allreduce_unique_id(ident), "allreduce_unique_id")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0, group=group)
        ident = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
        h = ctypes.c_void_p()
        cdt = _lib.ETP_BF16 if comm_dtype == torch.bfloat16 else _lib.ETP_F32
        with torch.cuda.device(device):
            _lib.check(self.L.etp_allreduce_init(ctypes.byref(h), ident, self.rank, self.world, cdt, int(max_bucket_elems)),
                       "allreduce_init")
        self.handle = h

    def ranks_seen(self) -> int:
        return int(self.L.etp_allreduce_world(self.handle))

    def bucket_ready(self, grads: torch.Tensor, start: int, end: int):
        self._lib.check(self.L.etp_allreduce_bucket_ready(self.handle, grads.data_ptr() + 4 * start, end - start,
                                                          torch.cuda.current_stream().cuda_stream), "allreduce_bucket_ready")

    def after(self, streams: Sequence[int]):
        """Order the communication stream after everything enqueued so far on the given raw stream handles."""
        cs = self.L.etp_allreduce_stream(self.handle)
        for st in streams:
            if st:
                self._lib.check(self.L.etp_stream_after(st, cs), "stream_after")

    def gather_rows(self, table: torch.Tensor, ids: torch.Tensor, capacity: int):
        """table [n_rows, row_len] fp32 view of the gradient arena; ids int64 on the device (this rank's touched rows)."""
        self._lib.check(self.L.etp_allreduce_gather_rows(self.handle, table.data_ptr(), table.shape[0], table.shape[1],
                                                         ids.data_ptr(), ids.numel(), int(capacity),
                                                         torch.cuda.current_stream().cuda_stream), "allreduce_gather_rows")

    def wait(self):
        self._lib.check(self.L.etp_allreduce_wait(self.handle, torch.cuda.current_stream().cuda_stream), "allreduce_wait")

    def _drain(self, timeout_s: float) -> bool:
        """Poll the communicator's stream from the host until it is idle; False when the deadline passes first."""
        import time
        t_end = time.monotonic() + timeout_s
        while not self.L.etp_allreduce_idle(self.handle):
            if time.monotonic() > t_end:
                return False
            time.sleep(0.002)
        return True

    def self_test(self, timeout_s: float = 60.0) -> Optional[str]:
        """Known-answer run of both collectives on this communicator.  Returns None when they completed within the deadline
        with the right values, else a description.  Dense: every rank contributes (rank + 1) -> mean (world + 1) / 2, on a
        length that exercises the reduce-scatter body AND the all-reduce tail of etp_allreduce_bucket_ready.  Row-sparse:
        rank r owns row r (value r + 1), all ranks share row `world` (value 10 * (rank + 1)), one id repeated."""
        W, r = self.world, self.rank
        dev = torch.device("cuda", torch.cuda.current_device())
        n = W * 64 * 3 + 17
        dense = torch.full((n,), float(r + 1), dtype=torch.float32, device=dev)
        table = torch.zeros(W + 2, 64, dtype=torch.float32, device=dev)
        table[r] = float(r + 1)
        table[W] = 10.0 * (r + 1)
        ids = torch.tensor([r, W, r], dtype=torch.int64, device=dev)
        torch.cuda.current_stream().synchronize()
        self.bucket_ready(dense, 0, n)
        self.gather_rows(table, ids, 4)
        if not self._drain(timeout_s):
            return f"collectives did not complete within {timeout_s:.0f} s"
        tol = 2e-2 if self.comm_dtype == torch.bfloat16 else 1e-6
        want = (W + 1) / 2.0
        bad = (dense - want).abs().max().item()
        if not bad <= tol * want:
            return f"dense mean off by {bad:.3g} (expected {want})"
        exp = torch.zeros_like(table)
        for k in range(W):
            exp[k] = (k + 1) / W
        exp[W] = 10.0 * want
        bad = (table - exp).abs().max().item()
        if not bad <= tol * 10.0 * want:
            return f"row-sparse mean off by {bad:.3g}"
        return None

    def abort(self):
        if getattr(self, "handle", None):
            self.L.etp_allreduce_abort(self.handle)
            self.close()

    def close(self):
        if getattr(self, "handle", None):
            self.L.etp_allreduce_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GradReducer:
    def __init__(self, flat_grads: torch.Tensor, ranges: Sequence[Tuple[int, int]], comm_dtype=torch.float32,
                 sparse_rows: Optional[Tuple[int, int, int]] = None, group=None, native: Optional[bool] = None):
        """ranges: element ranges [start, end) of the dense buckets, in the order they become ready.
        sparse_rows: (offset, n_rows, row_len) of a row-sparse table excluded from the dense ranges.
        native: use the library's communicator for the dense buckets (default: when the gradients are on a GPU, the group's
        backend is nccl/RCCL and ETP_DP_NATIVE != 0); False = torch.distributed collectives."""
        self.g = flat_grads
        self.ranges = list(ranges)
        self.comm_dtype = comm_dtype
        self.sparse = sparse_rows
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.native = None
        if native is None:
            native = (flat_grads.is_cuda and dist.is_initialized() and dist.get_backend(group) == "nccl"
                      and os.environ.get("ETP_DP_NATIVE", "1") != "0")
        if native and self.world > 1:
            # collective decision: either every rank gets the library communicator or every rank uses torch.distributed
            self.native = NativeComm.create(flat_grads.device, comm_dtype, max((e - s) for s, e in self.ranges), group)
        need_buf = comm_dtype != torch.float32 and self.native is None
        self._bufs = [torch.empty(e - s, dtype=comm_dtype, device=flat_grads.device) if need_buf else None
                      for s, e in self.ranges]
        self._pending: List = []

    @property
    def overlapped(self) -> bool:
        """True when reduce_bucket can order the communication after SIDE streams (the library communicator owns its stream);
        torch.distributed collectives only see the current stream, so their producers must be joined into it first."""
        return self.native is not None

    def reduce_bucket(self, i: int, async_op: bool = True, also: Sequence[int] = ()):
        """Start the mean-reduction of dense bucket i (call when its gradients are complete in stream order).  `also`: raw
        stream handles that hold producers of the bucket besides the current stream (weight-gradient / panorama streams of
        PlannerStep.run_data_parallel); only the library communicator can wait for them without stalling the current stream."""
        if self.world == 1:
            return
        s, e = self.ranges[i]
        if self.native is not None:
            self.native.after(also)
            self.native.bucket_ready(self.g, s, e)
            return
        if also:
            raise RuntimeError("torch.distributed collectives cannot wait for side streams: join them into the current stream "
                               "(PlannerStep.run_data_parallel(overlapped=False))")
        view = self.g[s:e]
        if self.comm_dtype == torch.float32:
            h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            self._pending.append((h, None, view))
        else:
            buf = self._bufs[i]
            buf.copy_(view)                       # fp32 -> bf16 on the producer's stream
            h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            self._pending.append((h, buf, view))

    def reduce_sparse_rows(self, row_ids: torch.Tensor, capacity: Optional[int] = None):
        """Row-sparse exchange for the word-embedding gradient: row_ids = this rank's touched rows (any order, may
        repeat).  Every rank contributes a block of `capacity` (id, row) slots, padded with a sentinel id whose row is
        zero, so `capacity` -- not len(row_ids) -- must be the same on every rank (the reference's collate pads to the
        per-batch maximum, tasks.py:322-364: the token count DIFFERS across ranks).  capacity=None takes the maximum of
        len(row_ids) over the ranks (one scalar all-reduce + a host read; pass B * max_txt_len to avoid the
        synchronisation).  Result: table gradient = mean over ranks, as the dense all-reduce would give."""
        if self.world == 1 or self.sparse is None:
            return
        off, n_rows, row_len = self.sparse
        table = self.g[off:off + n_rows * row_len].view(n_rows, row_len)
        ids = row_ids.reshape(-1).to(torch.long)
        if capacity is None:
            if self.native is not None:
                # the MAX below is a torch.distributed (second communicator) collective: dense buckets of the library communicator
                # may still be in flight on its private stream, and two RCCL communicators progressing concurrently can
                # deadlock -- drain ours first (this branch synchronises with the host anyway; pass `capacity` to avoid it)
                self.native.wait()
                torch.cuda.current_stream().synchronize()
            t = torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            capacity = int(t.item())
        if ids.numel() > capacity:
            raise ValueError(f"{ids.numel()} touched rows exceed the rank-independent capacity {capacity}")
        if self.native is not None:
            # the library communicator's own stream and communicator: ordered behind the dense buckets issued so far, no
            # second communicator in flight and no wait on the compute stream
            self.native.gather_rows(table, ids.contiguous(), capacity)
            return
        ids, _ = torch.sort(ids)
        first = torch.ones_like(ids, dtype=torch.bool)
        first[1:] = ids[1:] != ids[:-1]                                   # a repeated id contributes its row once
        rows = (table.index_select(0, ids) * first[:, None]).to(self.comm_dtype)
        table.index_fill_(0, ids, 0.0)                                    # own rows come back inside the gathered block
        pad = capacity - ids.numel()
        if pad:                                                           # sentinel slots: row 0 with an all-zero contribution
            ids = torch.cat([ids, ids.new_zeros(pad)])
            rows = torch.cat([rows, rows.new_zeros(pad, row_len)])
        all_ids = [torch.empty_like(ids) for _ in range(self.world)]
        all_rows = [torch.empty_like(rows) for _ in range(self.world)]
        dist.all_gather(all_ids, ids, group=self.group)
        dist.all_gather(all_rows, rows, group=self.group)
        inv = 1.0 / self.world
        table.index_add_(0, torch.cat(all_ids), torch.cat(all_rows).float() * inv)

    def finish(self):
        """Wait for the outstanding buckets (the consumer = torch's current stream) and, on the torch.distributed path,
        write the means back into the fp32 arena."""
        if self.native is not None:
            self.native.wait()
        for h, buf, view in self._pending:
            if h is not None:
                h.wait()
            if buf is None:
                view.mul_(1.0 / self.world)
            elif view.is_cuda:
                # one pass: bf16 sum -> fp32 mean (hand-written cast kernel through the C ABI)
                from . import _lib
                _lib.check(_lib.lib().etp_cast_bf16_to_f32(buf.data_ptr(), view.data_ptr(), view.numel(), 1.0 / self.world,
                                                           torch.cuda.current_stream().cuda_stream), "cast_bf16_to_f32")
            else:
                torch.mul(buf, 1.0 / self.world, out=view)
        self._pending.clear()

    def close(self):
        if self.native is not None:
            self.native.close()
            self.native = None


# ---- per-task static bucket sets (replaces DDP's find_unused_parameters=True, pretrain utils/misc.py:58) -------------------
# Which parameters a pre-training task touches is a static property of the task (pretrain_cmt.py:141-163 'mlm' vs :223-283
# 'sap'): the reference lets DDP discover the unused ones dynamically every step; here each task has a fixed list of arena
# ranges, and only those are reduced (the rest of the gradient arena is exactly zero on every rank).
_TASK_ONLY = {
    "sap": (".visn_self_att.", ".visn_inter.", ".visn_output.", "sprel_linear.", "global_sap_head."),
    "mlm": (".lang_self_att.", ".lang_inter.", ".lang_output.", "mlm_head."),
}


def task_uses_param(task: str, name: str) -> bool:
    """True when parameter `name` receives a gradient in pre-training task `task` ('sap' | 'mlm')."""
    if task not in _TASK_ONLY:
        raise ValueError(f"unknown pre-training task {task!r}")
    for t, pats in _TASK_ONLY.items():
        if t != task and any(p in name for p in pats):
            return False
    return True


def task_grad_ranges(model, task: str, max_gap: int = 0):
    """Element ranges [start, end) of the flat gradient arena that task `task` writes, merged over adjacent parameters
    (64-element alignment padding between two used parameters is bridged), in arena order, and the sparse word-table
    descriptor: SAP touches <= B*L rows of the word table (row-sparse exchange), MLM's tied decoder makes it dense (it is
    then part of the returned ranges and the descriptor is None)."""
    eng = model._engine
    spans = []
    word = None
    for name, shape, off in eng.table:
        n = 1
        for d in shape:
            n *= d
        if name == "embeddings.word_embeddings.weight":
            word = (off, shape[0], shape[1])
            if task != "mlm":
                continue
        if task_uses_param(task, name):
            spans.append((off, off + (n + 63) // 64 * 64))
    spans.sort()
    merged = []
    for s, e in spans:
        if merged and s - merged[-1][1] <= max_gap:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    total = eng.total
    return [(s, min(e, total)) for s, e in merged], (None if task == "mlm" else word)


def frozen_spans(model):
    """Arena spans [start, end) (64-element padded) of the parameters with requires_grad = False (fix_lang_embedding /
    fix_pano_embedding, vilmodel_cmt.py:675-682): DDP registers no hook for them (ss_trainer_ETP.py:208-212 wraps the module,
    whose frozen parameters never enter a bucket), so they must not travel here either."""
    eng = model._engine
    tab = {n: (off, shape) for n, shape, off in eng.table}
    spans = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            off, shape = tab[name]
            n = 1
            for d in shape:
                n *= d
            spans.append((off, min(eng.total, off + (n + 63) // 64 * 64)))
    spans.sort()
    merged = []
    for s, e in spans:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    return [(s, e) for s, e in merged]


def subtract_spans(ranges, spans):
    """ranges minus spans, order of `ranges` kept; a range cut in the middle becomes several consecutive ranges."""
    out = []
    for s, e in ranges:
        cur = s
        for fs, fe in spans:
            if fe <= cur or fs >= e:
                continue
            if fs > cur:
                out.append((cur, fs))
            cur = max(cur, fe)
        if cur < e:
            out.append((cur, e))
    return out


def planner_buckets_layered(model, text_groups: int = 3):
    """Finer buckets for overlap with the text-encoder backward: bucket 0 = non-text matrices (ready after the navigation
    and panorama backward), then one bucket per group of text layers in backward order (last layers first), then vectors
    and small tables.  Returns (ranges, sparse_word_table, [(layer_lo, layer_hi) per text bucket])."""
    eng = model._engine
    tab = {n: (off, shape) for n, shape, off in eng.table}
    n_l = int(eng.cconf.n_l)
    starts = [tab[f"lang_encoder.layer.{l}.attention.self.query.weight"][0] for l in range(n_l)]
    first_non_text = min(off for n, (off, _) in tab.items() if off < eng.n_matrix and not n.startswith("lang_encoder."))
    assert starts == sorted(starts) and (not starts or starts[0] == 0) and all(s < first_non_text for s in starts)
    starts.append(first_non_text)
    word_off, word_shape = tab["embeddings.word_embeddings.weight"]
    word_end = word_off + ((word_shape[0] * word_shape[1] + 63) // 64) * 64
    ranges, groups = [], []
    if first_non_text < eng.n_matrix:
        ranges.append((first_non_text, eng.n_matrix))
    frozen = frozen_spans(model)
    txt_frozen = n_l > 0 and subtract_spans([(starts[0], starts[n_l])], frozen) == []
    text_groups = max(1, min(text_groups, n_l)) if (n_l and not txt_frozen) else 0
    hi = n_l
    for gidx in range(text_groups):
        lo = (n_l * (text_groups - 1 - gidx)) // text_groups
        ranges.append((starts[lo], starts[hi]))
        groups.append((lo, hi))
        hi = lo
    # the text buckets keep their positions 1 .. len(groups) (PlannerStep.run_data_parallel announces bucket 1 + k after text
    # group k); frozen spans are cut out of bucket 0 and of the vector / table tail only
    tail = []
    if word_off > eng.n_matrix:
        tail.append((eng.n_matrix, word_off))
    if word_end < eng.total:
        tail.append((word_end, eng.total))
    word = (word_off, word_shape[0], word_shape[1])
    if frozen:
        head = subtract_spans(ranges[:1], frozen) if first_non_text < eng.n_matrix else []
        if first_non_text < eng.n_matrix and len(head) != 1:
            # bucket 0 must stay ONE range (its index is part of the step's announcement protocol): take the hull of what is left.
            # The frozen gaps inside the hull are reduced along with it: slots the backward never writes hold the zeros the arena
            # was allocated with; slots it does write although their parameter is frozen (fix_pano_embedding alone: the panorama
            # backward still produces the img_embeddings.* gradients on its way to token_type_embeddings(1)) carry real values.
            # Either way the optimizer's frozen bit drops them (etp_adamw_step mask values 2 / 3) -- wasted bytes, never an update.
            head = [(head[0][0], head[-1][1])] if head else [(first_non_text, first_non_text)]
        # `ranges[0]` is the non-text bucket only when there is one (first_non_text < n_matrix); without it the list starts with
        # text group 0, which must stay (the slice used to drop it)
        rest = ranges[1:] if first_non_text < eng.n_matrix else ranges
        ranges = head + rest + subtract_spans(tail, frozen)
        if subtract_spans([(word_off, word_end)], frozen) == []:
            word = None
        return ranges, word, groups
    return ranges + tail, word, groups


def planner_buckets(model, split_text: bool = True, dense_word_table: bool = False):
    """Dense bucket ranges for GlocalTextPathNavCMT's arena in backward-completion order, and the sparse word table.
    dense_word_table=True (the pre-training MLM task: the tied decoder makes the word-embedding gradient dense) reduces the
    table as one more dense range and returns None for the sparse descriptor.

    Arena = [GEMM matrices (text layers first, then pano / x-layers / head) | vectors | embedding tables].
    Backward order of a step: navigation -> panorama -> text, so bucket 0 = non-text matrices (ready after the
    panorama backward), bucket 1 = text matrices + vectors + small tables (ready at the end)."""
    eng = model._engine
    tab = {n: (off, shape) for n, shape, off in eng.table}
    word_off, word_shape = tab["embeddings.word_embeddings.weight"]
    word_end = word_off + ((word_shape[0] * word_shape[1] + 63) // 64) * 64
    first_non_text = min(off for n, (off, _) in tab.items() if off < eng.n_matrix and not n.startswith("lang_encoder."))
    assert all(off < first_non_text for n, (off, _) in tab.items() if n.startswith("lang_encoder.") and off < eng.n_matrix)
    ranges = []
    if split_text and first_non_text < eng.n_matrix:
        ranges.append((first_non_text, eng.n_matrix))
        ranges.append((0, first_non_text))
    else:
        ranges.append((0, eng.n_matrix))
    # vectors + tables other than the word table
    if word_off > eng.n_matrix:
        ranges.append((eng.n_matrix, word_off))
    if word_end < eng.total:
        ranges.append((word_end, eng.total))
    word = (word_off, word_shape[0], word_shape[1])
    frozen = frozen_spans(model)
    if dense_word_table:
        ranges.append((word_off, word_end))
        word = None
    if frozen:
        ranges = subtract_spans(ranges, frozen)
        if word is not None and subtract_spans([(word_off, word_end)], frozen) == []:
            word = None
    return ranges, word

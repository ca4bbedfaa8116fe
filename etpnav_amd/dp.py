"""Data-parallel gradient averaging over RCCL/xGMI (one process per GPU, torch.distributed backend "nccl" = RCCL).

Replaces ``DistributedDataParallel(self.policy.net, ...)`` of ss_trainer_ETP.py:208-212 (and pretrain
utils/misc.py:52-65) for the planner: episodes shard across ranks with no activation exchange; the only collective
is the mean of the parameter gradients, once per step.

MI355X-first choices:
  * all gradients already live in ONE flat fp32 arena, so the reduction runs over a few large contiguous buckets
    (no per-parameter bucketing / flattening copies);
  * gradients travel as bf16 (half the xGMI bytes; accumulation stays fp32 locally) unless ``comm_dtype=float32``;
  * the word-embedding gradient is row-sparse (<= B*L of 30 522 / 250 002 rows are non-zero): instead of all-reducing
    the dense 94 MB / 768 MB table, ranks all-gather their (row ids, rows) and scatter-add locally — same result as the
    dense mean up to summation order;
  * buckets are issued on RCCL's stream right after the backward segment that completes them, so the text-encoder
    backward overlaps the reduction of everything computed before it (see ``PlannerStep``/bench.py).
Works unchanged on CPU with the ``gloo`` backend (tests/test_dp_gloo.py, world_size 2).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat_grads: torch.Tensor, ranges: Sequence[Tuple[int, int]], comm_dtype=torch.bfloat16,
                 sparse_rows: Optional[Tuple[int, int, int]] = None, group=None):
        """ranges: element ranges [start, end) of the dense buckets, in the order they become ready.
        sparse_rows: (offset, n_rows, row_len) of a row-sparse table excluded from the dense ranges."""
        self.g = flat_grads
        self.ranges = list(ranges)
        self.comm_dtype = comm_dtype
        self.sparse = sparse_rows
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._bufs = [torch.empty(e - s, dtype=comm_dtype, device=flat_grads.device) if comm_dtype != torch.float32 else None
                      for s, e in self.ranges]
        self._pending: List = []

    def reduce_bucket(self, i: int, async_op: bool = True):
        """Start the mean-reduction of dense bucket i (call when its gradients are complete)."""
        if self.world == 1:
            return
        s, e = self.ranges[i]
        view = self.g[s:e]
        if self.comm_dtype == torch.float32:
            h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            self._pending.append((h, None, view))
        else:
            buf = self._bufs[i]
            buf.copy_(view)                       # fp32 -> bf16 on the producer's stream
            h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            self._pending.append((h, buf, view))

    def reduce_sparse_rows(self, row_ids: torch.Tensor):
        """Row-sparse exchange for the word-embedding gradient: row_ids = this rank's touched rows (any order, may
        repeat).  Result: table gradient = mean over ranks, as the dense all-reduce would give."""
        if self.world == 1 or self.sparse is None:
            return
        off, n_rows, row_len = self.sparse
        table = self.g[off:off + n_rows * row_len].view(n_rows, row_len)
        ids = torch.unique(row_ids.reshape(-1))
        # fixed-size exchange: pad to the max count over ranks (B*L is equal across ranks for synthetic batches)
        cnt = torch.tensor([ids.numel()], device=ids.device, dtype=torch.long)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt, group=self.group)
        m = int(max(c.item() for c in cnts))
        pad_ids = torch.zeros(m, dtype=torch.long, device=ids.device); pad_ids[:ids.numel()] = ids
        rows = torch.zeros(m, row_len, dtype=self.comm_dtype, device=ids.device)
        rows[:ids.numel()] = table[ids].to(self.comm_dtype)
        all_ids = [torch.empty_like(pad_ids) for _ in range(self.world)]
        all_rows = [torch.empty_like(rows) for _ in range(self.world)]
        dist.all_gather(all_ids, pad_ids, group=self.group)
        dist.all_gather(all_rows, rows, group=self.group)
        table[ids] = 0
        for r in range(self.world):
            k = int(cnts[r].item())
            table.index_add_(0, all_ids[r][:k], all_rows[r][:k].float())
        table.mul_(1.0 / self.world)

    def finish(self):
        """Wait for the outstanding buckets and write the means back into the fp32 arena."""
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
    text_groups = max(1, min(text_groups, n_l)) if n_l else 0
    hi = n_l
    for gidx in range(text_groups):
        lo = (n_l * (text_groups - 1 - gidx)) // text_groups
        ranges.append((starts[lo], starts[hi]))
        groups.append((lo, hi))
        hi = lo
    if word_off > eng.n_matrix:
        ranges.append((eng.n_matrix, word_off))
    if word_end < eng.total:
        ranges.append((word_end, eng.total))
    return ranges, (word_off, word_shape[0], word_shape[1]), groups


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
    if dense_word_table:
        ranges.append((word_off, word_end))
        return ranges, None
    return ranges, (word_off, word_shape[0], word_shape[1])

"""Multi-GPU sharding of the candidate batch: one process per GPU, candidates
partitioned by rank, one tiny exchange per plan iteration.

The reference has no distributed path at all (single process, std::thread pool);
this is the MI355X-native replacement: `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests) carries
  * all-gather of each rank's k best (return, global index) pairs -> global top-k,
  * broadcast of the winner's spline values from its owner rank.
Payloads are a few hundred bytes, i.e. latency-bound; no bandwidth tuning applies.
"""
from __future__ import annotations

import numpy as np


class RankGroup:
    def __init__(self, dist, device=None):
        """`dist` = torch.distributed (already initialised); `device` = torch device for
        collective tensors (cuda:<local_rank> with RCCL, cpu with gloo)."""
        import torch
        self.torch = torch
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.device = device if device is not None else torch.device("cpu")

    def barrier(self):
        self.dist.barrier()

    def owner_of(self, global_idx, num_trajectory):
        """rank whose contiguous candidate range holds `global_idx` (the first N % world ranks hold one more candidate)"""
        q, r = divmod(num_trajectory, self.world)
        if global_idx >= num_trajectory:
            return self.world - 1
        return global_idx // (q + 1) if global_idx < r * (q + 1) else r + (global_idx - r * (q + 1)) // max(q, 1)

    def merge_topk(self, idx, ret, k):
        """Global k best from each rank's local k best; ties broken by global index."""
        t = self.torch
        kk = len(idx)
        local = t.full((k, 2), float("inf"), dtype=t.float64, device=self.device)
        local[:kk, 0] = t.as_tensor(np.asarray(ret, np.float64), device=self.device)
        local[:kk, 1] = t.as_tensor(np.asarray(idx, np.float64), device=self.device)
        local[kk:, 1] = float(2 ** 52)
        gathered = [t.empty_like(local) for _ in range(self.world)]
        self.dist.all_gather(gathered, local)
        allp = t.cat(gathered).cpu().numpy()
        order = np.lexsort((allp[:, 1], allp[:, 0]))[:k]
        keep = allp[order]
        keep = keep[keep[:, 1] < 2 ** 52]
        return keep[:, 1].astype(np.int64), keep[:, 0]

    def exchange_best(self, idx, best_ret, nominal_ret, values):
        """Predictive Sampling exchange: all-gather (best return, global index, nominal return) and
        broadcast the winner's spline values from its owner. Ties go to the lowest global index."""
        t = self.torch
        local = t.tensor([best_ret, float(idx), nominal_ret], dtype=t.float64, device=self.device)
        gathered = [t.empty_like(local) for _ in range(self.world)]
        self.dist.all_gather(gathered, local)
        allp = t.stack(gathered).cpu().numpy()
        rets = np.where(np.isnan(allp[:, 0]), np.inf, allp[:, 0])
        owner = int(np.lexsort((allp[:, 1], rets))[0])
        nominal = float(allp[0, 2])                       # global candidate 0 lives on rank 0
        vals = self.broadcast_array(values if self.rank == owner else None, np.asarray(values).shape, src=owner)
        return int(allp[owner, 1]), float(allp[owner, 0]), nominal, vals

    def broadcast_scalar(self, value, src=0):
        t = self.torch
        x = t.tensor([0.0 if value is None else float(value)], dtype=t.float64, device=self.device)
        self.dist.broadcast(x, src=src)
        return float(x.item())

    def broadcast_array(self, values, shape, src=0):
        t = self.torch
        if values is None:
            x = t.zeros(shape, dtype=t.float64, device=self.device)
        else:
            x = t.as_tensor(np.ascontiguousarray(values, np.float64).reshape(shape), device=self.device)
        self.dist.broadcast(x, src=src)
        return x.cpu().numpy()

    def time_shard(self, T):
        """[begin, end) of this rank's share of T time steps (contiguous, the first T % world ranks get one more)"""
        q, r = divmod(T, self.world)
        begin = self.rank * q + min(self.rank, r)
        return begin, begin + q + (1 if self.rank < r else 0)

    def sharded_transition_fd(self, fd, times, states, actions, **kw):
        """iLQG derivative sweep over ranks: every (t, perturbation column) of ModelDerivatives::Compute
        is independent, so rank r evaluates `fd(times[b:e], states[b:e], actions[b:e], **kw)` -> (A, B, C, D) for its share of
        the time steps (fd = capi.Context.transition_fd) and the blocks are all-gathered; every rank returns the full
        (T, ...) arrays, bit-identical to the unsharded call. The backward pass is a serial recursion and stays replicated.
        (At the A1's T = 36 the sweep takes 1.4 ms on one GPU: sharding it pays only for long horizons.)"""
        t = self.torch
        T = len(times)
        b, e = self.time_shard(T)
        states, actions = np.asarray(states, np.float64).reshape(T, -1), np.asarray(actions, np.float64).reshape(T, -1)
        parts = fd(np.asarray(times, np.float64)[b:e], states[b:e], actions[b:e], **kw) if e > b else None
        shapes = None
        if parts is not None:
            shapes = [p.shape[1:] for p in parts]
        # per-step sizes are the same on every rank; a rank with an empty share learns them from rank 0 (which is never
        # empty when T >= 1)
        meta = t.zeros(8, dtype=t.int64, device=self.device)
        if self.rank == 0:
            flat = [d for sh in shapes for d in sh]
            meta[:len(flat)] = t.as_tensor(flat, dtype=t.int64, device=self.device)
        self.dist.broadcast(meta, src=0)
        dims = meta.cpu().numpy().tolist()
        shapes = [(dims[0], dims[1]), (dims[2], dims[3]), (dims[4], dims[5]), (dims[6], dims[7])]
        per_step = sum(a * c for a, c in shapes)
        cap = -(-T // self.world)  # equal-sized all-gather blocks: pad the smaller shares
        local = t.zeros(cap * per_step, dtype=t.float64, device=self.device)
        if parts is not None:
            packed = np.concatenate([np.asarray(p, np.float64).reshape(e - b, -1) for p in parts], axis=1).reshape(-1)
            local[:packed.size] = t.as_tensor(packed, device=self.device)
        gathered = [t.empty_like(local) for _ in range(self.world)]
        self.dist.all_gather(gathered, local)
        rows = []
        q, r = divmod(T, self.world)
        for k, g in enumerate(gathered):
            n = q + (1 if k < r else 0)
            rows.append(g.cpu().numpy()[:n * per_step].reshape(n, per_step))
        full = np.concatenate(rows, axis=0)
        out, o = [], 0
        for a, c in shapes:
            out.append(full[:, o:o + a * c].reshape(T, a, c).copy())
            o += a * c
        return tuple(out)

    def sum_array(self, values):
        """all-reduce(sum) of a small fp64 vector (the CE elite moments)."""
        t = self.torch
        x = t.as_tensor(np.ascontiguousarray(values, np.float64), device=self.device).clone()
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
        return x.cpu().numpy()

    def max_scalar(self, value):
        t = self.torch
        x = t.tensor([float(value)], dtype=t.float64, device=self.device)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX)
        return float(x.item())

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
        n = num_trajectory // self.world
        return min(global_idx // n, self.world - 1) if n > 0 else self.world - 1

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

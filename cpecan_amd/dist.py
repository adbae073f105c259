"""Multi-GPU plumbing for the hot path: one process per GPU, pairs sharded statically,
no data-path collective for posteriors; the EM expectation step adds ONE all-reduce(SUM) of the count vector
(transitions S*S, emissions S*16, likelihood) -- the GPU counterpart of cPecanEm.py:184-188 summing per-shard
expectation files.  torch.distributed backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
import ctypes as C

import numpy as np

from . import api


def shard_bounds(n_items, rank, world_size):
    """Contiguous static shard [lo, hi) of n_items for this rank; shards differ in size by at most one."""
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def hmm_to_vector(hmm):
    """(transitions[S*S], emissions[S*16], likelihood) as one float64 vector."""
    S = hmm.stateNumber
    return np.concatenate([np.array(hmm.transitions[:S * S], dtype=np.float64),
                           np.array(hmm.emissions[:S * 16], dtype=np.float64),
                           np.array([hmm.likelihood], dtype=np.float64)])


def vector_to_hmm(vec, hmm):
    S = hmm.stateNumber
    for i in range(S * S):
        hmm.transitions[i] = float(vec[i])
    for i in range(S * 16):
        hmm.emissions[i] = float(vec[S * S + i])
    hmm.likelihood = float(vec[S * S + S * 16])
    return hmm


def local_device_index():
    """The GPU this rank works on: torch's current device when a GPU is visible (launchers set it from LOCAL_RANK),
    else LOCAL_RANK, else 0."""
    import os
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:  # noqa: BLE001 -- no torch / no GPU: fall through
        pass
    return int(os.environ.get("LOCAL_RANK", "0"))


def reduce_device_for_backend():
    """Where the count vector must live for the collective: RCCL ("nccl") reduces GPU tensors only; gloo takes CPU ones."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", local_device_index())
    return torch.device("cpu")


def allreduce_hmm(hmm, device=None):
    """In-place sum of expectation counts over all ranks (one collective of S*S + S*16 + 1 doubles).  `device` defaults
    to what the process group's backend needs (a GPU tensor under RCCL)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return hmm
    t = torch.from_numpy(hmm_to_vector(hmm)).to(device if device is not None else reduce_device_for_backend())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return vector_to_hmm(t.cpu().numpy(), hmm)


def expectation_step(sM, problems, p, pseudo=1e-12, device_index=None, reduce_device=None):
    """One E-step over this rank's problems on its GPU (default: this rank's own, see local_device_index), then the
    all-reduce; returns the summed (un-normalised) Hmm.  problems: iterable of (sX, sY, anchors, raggedLeft, raggedRight)."""
    if device_index is None:
        device_index = local_device_index()
    acc = api.hmm_constructEmpty(0.0, sM.type)
    with api.Batch(sM, p, emit=api.EMIT_EXPECT, device=device_index) as b:
        n = 0
        for sx, sy, anchors, rl, rr in problems:
            b.add(sx, sy, anchors, rl, rr)
            n += 1
        if n:
            b.upload()
            b.run()
            b.download()
            b.expectations(acc)
    allreduce_hmm(acc, reduce_device)
    # the reference seeds every shard's Hmm with the pseudo count (cPecanRealign.c:493); add it once globally
    S = acc.stateNumber
    for i in range(S * S):
        acc.transitions[i] += pseudo
    for i in range(S * 16):
        acc.emissions[i] += pseudo
    return acc


def cost_balanced_bounds(costs, rank, world_size):
    """Contiguous shard [lo, hi) of items with the given costs such that every rank gets about the same total cost
    (cut points at the multiples of total / world_size in the running sum).  Contiguous, so outputs concatenate in input
    order; with equal costs it reduces to shard_bounds up to rounding."""
    c = np.asarray(costs, dtype=np.float64)
    if len(c) == 0:
        return 0, 0
    run = np.concatenate([[0.0], np.cumsum(np.maximum(c, 0.0) + 1e-9)])
    cuts = [int(np.searchsorted(run, run[-1] * k / world_size, side="left")) for k in range(world_size + 1)]
    cuts[0], cuts[-1] = 0, len(c)
    for k in range(1, world_size + 1):
        cuts[k] = max(cuts[k], cuts[k - 1])
    return cuts[rank], cuts[rank + 1]


def lpt_assign(costs, world_size):
    """Longest-processing-time-first assignment of items to ranks: items by decreasing cost, each to the
    least loaded rank so far; ties by index, so every rank computes the same answer without talking to the others.
    Returns one sorted index array per rank.  Equal costs give a round-robin deal."""
    import heapq
    c = np.asarray(costs, dtype=np.float64)
    order = np.lexsort((np.arange(len(c)), -c))
    heap = [(0.0, r) for r in range(world_size)]
    out = [[] for _ in range(world_size)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(int(i))
        heapq.heappush(heap, (load + float(c[i]), r))
    return [np.array(sorted(v), dtype=np.int64) for v in out]


def cigar_cost(c, expansion=4):
    """Band cells a cigar costs the aligner, near enough: diagonals times band width."""
    return (abs(c.end1 - c.start1) + abs(c.end2 - c.start2) + 1) * (expansion + 1)


def realign_sharded(cigars, realign_fn, expansion=4):
    """BASELINE config 4 across GPUs: every rank realigns one contiguous, cost-balanced shard of the cigars with
    realign_fn (its own Realigner, bound to its own GPU) -- no data-path collective -- and rank 0 receives the realigned
    cigars of all ranks in input order (a gather of the result objects; None on the other ranks)."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = cost_balanced_bounds([cigar_cost(c, expansion) for c in cigars], rank, world)
    mine = realign_fn(cigars[lo:hi])
    if world == 1:
        return mine
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    return [c for part in parts for c in part] if rank == 0 else None


def realign_expectations_sharded(cigars, expectations_fn, hmm, expansion=4, reduce_device=None):
    """--outputExpectations across GPUs: per-rank counts of a contiguous shard, then the one all-reduce of the EM step."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = cost_balanced_bounds([cigar_cost(c, expansion) for c in cigars], rank, world)
    expectations_fn(cigars[lo:hi], hmm)
    return allreduce_hmm(hmm, reduce_device)

"""world_size-2 gloo tests (CPU) of the N>1 path: static sharding of pairs and the EM count all-reduce.
Per-shard expectation counts come from the oracle here (no GPU in this container); the code under test is the
product's sharding and collective plumbing (cpecan_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from cpecan_amd import api, dist as cdist
from cpecan_amd.workload import make_batch


def test_shard_bounds_cover_without_overlap():
    for n in (0, 1, 7, 8, 9, 10000):
        for w in (1, 2, 3, 8):
            spans = [cdist.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b - a >= d - c >= b - a - 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    problems = make_batch(5, 12, 150, 10)
    lo, hi = cdist.shard_bounds(len(problems), rank, world)
    om, op = ob.model(ob.FIVE_STATE), ob.params(diagonalExpansion=10)
    oacc = ob.hmm(ob.FIVE_STATE, 0.0)
    for sx, sy, a in problems[lo:hi]:
        ob.expectations(om, oacc, sx, sy, a, op, True, True)
    # hand the shard's counts to the product's Hmm type and all-reduce them
    h = api.hmm_constructEmpty(0.0, api.fiveState)
    for i in range(25):
        h.transitions[i] = oacc.T[i]
    for i in range(80):
        h.emissions[i] = oacc.E[i]
    h.likelihood = oacc.likelihood
    cdist.allreduce_hmm(h)
    if rank == 0:
        np.save(out_path, cdist.hmm_to_vector(h))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_of_expectation_counts(tmp_path):
    out = str(tmp_path / "sum.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    problems = make_batch(5, 12, 150, 10)
    om, op = ob.model(ob.FIVE_STATE), ob.params(diagonalExpansion=10)
    oacc = ob.hmm(ob.FIVE_STATE, 0.0)
    for sx, sy, a in problems:
        ob.expectations(om, oacc, sx, sy, a, op, True, True)
    want = np.concatenate([np.array(oacc.T[:25]), np.array(oacc.E[:80]), [oacc.likelihood]])
    assert np.allclose(got, want, rtol=1e-12, atol=0)


def test_hmm_vector_round_trip():
    h = api.hmm_constructEmpty(0.0, api.threeState)
    for i in range(9):
        h.transitions[i] = i + 0.5
    for i in range(48):
        h.emissions[i] = 100 + i
    h.likelihood = -12.25
    v = cdist.hmm_to_vector(h)
    assert v.shape == (58,)
    g = cdist.vector_to_hmm(v, api.hmm_constructEmpty(0.0, api.threeState))
    assert list(g.transitions)[:9] == list(h.transitions)[:9] and g.likelihood == h.likelihood


def test_cost_balanced_bounds():
    import random
    rng = random.Random(3)
    for _ in range(200):
        n, w = rng.randrange(0, 300), rng.choice([1, 2, 3, 8])
        costs = [rng.choice([1, 5, 100, 2000]) for _ in range(n)]
        spans = [cdist.cost_balanced_bounds(costs, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(b == c for (_, b), (c, _) in zip(spans, spans[1:])) and all(a <= b for a, b in spans)
        if n:
            share = sum(costs) / w
            for a, b in spans:  # no shard exceeds its share by more than one item
                assert sum(costs[a:b]) <= share + max(costs) + 1e-6
    assert [cdist.cost_balanced_bounds([1] * 10, r, 2) for r in range(2)] == [(0, 5), (5, 10)]


def _realign_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpecan_amd.realign import Cigar
    cigars = [Cigar("t", 0, 10 * (i + 1), True, "q", 0, 10 * (i + 1), True, float(i), [(0, 10 * (i + 1))]) for i in range(23)]

    def fake_realign(shard):  # stands in for Realigner.realign on this rank's GPU: tags every cigar with the rank
        return [Cigar(c.contig1, c.start1, c.end1, True, c.contig2, c.start2, c.end2, True, c.score + 1000 * (rank + 1), c.ops)
                for c in shard]

    got = cdist.realign_sharded(cigars, fake_realign)
    if rank == 0:
        np.save(out_path, np.array([c.score for c in got]))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_realign_sharding_keeps_input_order(tmp_path):
    out = str(tmp_path / "scores.npy")
    mp.spawn(_realign_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    scores = np.load(out)
    assert len(scores) == 23
    assert [int(s) % 1000 for s in scores] == list(range(23))  # input order kept
    ranks = [int(s) // 1000 for s in scores]
    assert ranks == sorted(ranks) and set(ranks) == {1, 2}  # contiguous shards, both ranks worked
    cut = ranks.index(2)
    cost = [10 * (i + 1) for i in range(23)]
    assert abs(sum(cost[:cut]) - sum(cost[cut:])) <= max(cost) * 2  # balanced by cost, not by count
    assert cut > 23 // 2


def test_lpt_assign_deals_every_item_once_and_balances():
    """longest-first bin packing of pairs by band cells.  Every rank computes the same deal on its own."""
    from cpecan_amd import workload
    costs = workload.pair_costs("4", 3000)  # mixed lengths, 100-5000 bp
    for world in (1, 2, 3, 8):
        parts = cdist.lpt_assign(costs, world)
        allidx = np.concatenate(parts)
        assert sorted(allidx.tolist()) == list(range(3000))
        loads = np.array([costs[p].sum() for p in parts])
        assert loads.max() - loads.min() <= costs.max()          # the LPT bound: within one item of each other
        assert [p.tolist() for p in cdist.lpt_assign(costs, world)] == [p.tolist() for p in parts]  # deterministic
    # equal costs: a round-robin deal
    assert [p.tolist() for p in cdist.lpt_assign(np.ones(7), 3)] == [[0, 3, 6], [1, 4], [2, 5]]


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (before anything touches a GPU),
    deals the batch out (strong scaling is the default for N > 1), runs the rendezvous and the reductions.  Here on the
    CPU with gloo and --dry-run (no GPU work); the N = 1 form of the same command is what the driver runs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo",
                          "--pairs", "301", "--config", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["world_size"] == 2
    assert line["config"]["pairs_total"] == 301                   # every pair on exactly one rank
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run", "--pairs", "301",
                          "--config", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert line1["config"]["estimated_cells_total"] == line["config"]["estimated_cells_total"]  # the SAME batch


def test_reduce_device_follows_the_backend():
    """under RCCL the count vector must be a GPU tensor; under gloo a CPU one.  (The gloo half runs here.)"""
    port = _free_port()

    def _body():
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=0, world_size=1)
        try:
            assert cdist.reduce_device_for_backend() == torch.device("cpu")
        finally:
            dist.destroy_process_group()
    _body()
    assert cdist.local_device_index() == int(os.environ.get("LOCAL_RANK", "0"))  # no GPU here: LOCAL_RANK, else 0

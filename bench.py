#!/usr/bin/env python3
"""bench.py -- band DP cells/s of the banded pair-HMM forward+backward+posterior path on MI355X.

A "step" is one pass of the hot path (one launch of the fused sweep kernel) over one batch of synthetic
sequence pairs that is already resident in HBM.  Default workload = BASELINE.json configs[2] ("B"):
10 000 pairs, 2 kb x 2 kb, stateMachine5, diagonalExpansion (band) 100, anchors every 50 bp.
With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank runs its own 10 000 pairs
(weak scaling, no data-path collective); value = cells of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def measured_traffic(config, n_pairs):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this very command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  Counters cannot be read from inside bench.py, so the figure is only quoted for the exact workload it
    was measured on (config B, 10 000 pairs); otherwise null."""
    import glob
    if config != "B" or n_pairs != 10000:
        return None
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")),
                   key=lambda f: [int(t) for t in re.findall(r"\d+", os.path.basename(f))])  # r01_v9 < r01_v10
    if not files:
        return None
    try:
        return json.load(open(files[-1]))["hbm_bytes_per_launch"]
    except Exception:
        return None


def build_batch(api, workload, cfg, n_pairs, first, device):
    mtype = api.fiveState if cfg["model"] == "fiveState" else api.threeState
    sm = api.stateMachine5_construct(mtype) if mtype == api.fiveState else api.stateMachine3_construct(mtype)
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=cfg["expansion"],
                                                         splitMatrixBiggerThanThis=10 ** 15)
    problems = workload.make_batch(cfg["seed"], n_pairs, cfg["length"], cfg["expansion"], first=first)
    b = api.Batch(sm, p, device=device)
    for sx, sy, a in problems:
        b.add(sx, sy, a if cfg["anchors"] else ())
    t0 = time.time()
    b.upload()
    return b, problems, p, mtype, time.time() - t0


def cpu_baseline(cfg, problems, threads):
    """The CPU oracle (a port of the reference algorithm, see oracle/) on a bounded sample, host cores only."""
    import oracle_binding as ob
    mtype = ob.FIVE_STATE if cfg["model"] == "fiveState" else ob.THREE_STATE
    op = ob.params(diagonalExpansion=cfg["expansion"], splitMatrixBiggerThanThis=10 ** 15)
    sample = problems if cfg["anchors"] else [(sx, sy, ()) for sx, sy, _ in problems]
    t0 = time.time()
    _, cells = ob.batch_aligned_pairs(ob.model(mtype), sample, op, threads=threads)
    dt = time.time() - t0
    return cells / dt, cells, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="B", choices=["A", "B", "plumbing"])
    ap.add_argument("--pairs", type=int, default=0, help="override the number of pairs per GPU (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from cpecan_amd import api, workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available() or api.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    torch.zeros(1, device="cuda")  # create the HIP context now: a once-per-process cost, not part of any upload
    torch.cuda.synchronize()
    cfg = dict(workload.CONFIGS[args.config])
    n_pairs = args.pairs or cfg["n_pairs"]
    # weak scaling: rank r aligns pairs [r*n, (r+1)*n) of the same seeded stream -- independent objects, no exchange
    batch, problems, params, mtype, upload_s = build_batch(api, workload, cfg, n_pairs, rank * n_pairs, local_rank)
    st = batch.stats()
    cells = st.cells
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run(stream.cuda_stream)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        batch.run(stream.cuda_stream)
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(1, args.steps)  # HIP events on the launch stream: avg launch duration

    total_cells = cells
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([cells], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = int(c.item())

    # results back on the host once (not part of the timed region): PCIe + list assembly, and a parity spot check
    t1 = time.perf_counter()
    batch.download()
    d2h_s = time.perf_counter() - t1
    st = batch.stats()

    if rank == 0:
        S = 5 if cfg["model"] == "fiveState" else 3
        bytes_per_cell = 16 * S + 8  # F written once + read once (2*S*8 B) + one posterior word
        achieved = cells * bytes_per_cell / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "band DP cells/s (banded fwd+bwd+posterior)",
            "value": total_cells * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "config %s: %d pairs/GPU, %d x ~%d bp, %s, diagonalExpansion=%d, anchors every 50 bp, "
                            "threshold=0.01, traceback 1000/40" % (args.config, n_pairs, cfg["length"], cfg["length"],
                                                                   cfg["model"], cfg["expansion"]),
                "pairs_per_gpu": n_pairs,
                "cells_per_gpu": cells,
                "parallelism": "pairs sharded over %d GPU(s), no collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "cpecan_pairhmm_sweep<%d, true, 0>" % S,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.config, n_pairs),
                "bytes_per_cell": bytes_per_cell,
                "kernel_ms": kernel_ms,
            },
            "e2e": {"upload_s": upload_s, "download_and_assemble_s": d2h_s, "pairs_emitted": int(st.pairs),
                    "device_bytes": int(st.deviceBytes), "waves": int(st.wavesPerLaunch)},
        }
        if not args.no_cpu_baseline and world == 1:
            # (rank 0, N=1 only) the GPU box gives one GPU a 16-core share of its host CPUs; do not oversubscribe it
            threads = int(os.environ.get("CPECAN_BENCH_CPU_THREADS", min(os.cpu_count() or 1, 16)))
            sample = problems[:min(len(problems), 64 * threads)]
            v, ccells, dt = cpu_baseline(cfg, sample, threads)
            out["cpu_baseline"] = {
                "value": v, "unit": "cells/s", "cores": threads, "kind": "port",
                "sample": "first %d pairs of the same workload (%d cells), oracle C restatement, OpenMP over pairs, "
                          "%.1f s" % (len(sample), ccells, dt),
            }
            # parity spot check of the timed batch against the oracle (first 4 pairs)
            import oracle_binding as ob
            from parity import assert_pairs_match
            om = ob.model(ob.FIVE_STATE if cfg["model"] == "fiveState" else ob.THREE_STATE)
            op = ob.params(diagonalExpansion=cfg["expansion"], splitMatrixBiggerThanThis=10 ** 15)
            for i in range(min(4, len(problems))):
                sx, sy, a = problems[i]
                want = ob.aligned_pairs(om, sx, sy, a if cfg["anchors"] else (), op)
                assert_pairs_match(batch.result(i), want, threshold=params.threshold)
            out["parity_spot_check"] = "4 pairs match the oracle"
        print(json.dumps(out), flush=True)
    batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

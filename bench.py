#!/usr/bin/env python3
"""bench.py -- band DP cells/s of the banded pair-HMM forward+backward+posterior path on MI355X.

A "step" is one pass of the hot path (one run of the batch's sweep kernels) over one batch of synthetic sequence
pairs that is already resident in HBM.  Default workload = BASELINE.json configs[2] ("B"): 10 000 pairs,
2 kb x 2 kb, stateMachine5, diagonalExpansion (band) 100, anchors every 50 bp.  --config A | 4 | 5 | plumbing select
the other BASELINE configs (4: cPecanRealign mode, 50 000 mixed-length pairs, expansion 4, split at 10, ragged ends;
5: the EM expectation step on 100 000 pairs, its count all-reduce inside the step when N > 1).

--gpus N: one process per GPU.  Started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment)
the process is one rank; started plainly with N > 1 it launches the N ranks itself -- as child processes, before
anything touches the GPU -- and exits with their code.  --scaling strong (default for N > 1): the config's pairs are
dealt out to the ranks longest-first by band cells, no data-path collective; --scaling weak: every rank
runs the whole config's number of pairs.  value = cells of all ranks / max-over-ranks time.

Besides the kernel-only `value` the line carries `value_e2e` (SURVEY 8d's wall clock: problems in host memory ->
result lists in host memory; steady state of a pipeline two batches deep, every stage of every batch inside the clock)
and `e2e` with the unpipelined stages and the pipeline's fill and drain.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_SOURCES = ["cpecan_kernels.hip", "cpk_device_common.inl", "cpk_sweep.inl", "cpk_team.inl", "cpk_packed.inl",
                  "cpk_table_gather.inl", "cpk_post.inl", "cpk_cells.inl", "cpecan_band.inl", "cpecan_internal.h"]


def kernel_source_hash():
    """Identifies the device code a profile was taken on: sha256 over the HIP translation unit's files."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "cpecan_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_profile(kind, config, n_pairs):
    """A figure that cannot be measured from inside bench.py (PMC counters need rocprofv3 around the process): read from
    the newest profiles/r*_<kind>.json whose recorded kernel-source hash, config and pair count are THIS build's and this
    run's; otherwise None (a stale number is worse than none)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % kind)),
                   key=lambda f: [int(t) for t in re.findall(r"\d+", os.path.basename(f))])
    src = kernel_source_hash()
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        if d.get("kernel_source_hash") == src and d.get("config") == config and d.get("pairs") == n_pairs:
            return d
    return None


def self_launch(args):
    """python bench.py --gpus N without a launcher: start the N ranks as children (nothing here has touched the GPU)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def model_and_params(api, cfg):
    mtype = api.fiveState if cfg["model"] == "fiveState" else api.threeState
    sm = api.stateMachine5_construct(mtype) if mtype == api.fiveState else api.stateMachine3_construct(mtype)
    p = api.pairwiseAlignmentBandingParameters_construct(diagonalExpansion=cfg["expansion"],
                                                         splitMatrixBiggerThanThis=cfg.get("split", 10 ** 15))
    return sm, p, mtype


def cpu_baseline(cfg, problems, threads, reps=3):
    """The CPU oracle (a port of the reference algorithm, see oracle/) on a bounded sample of the same workload; median
    of `reps` repetitions."""
    import oracle_binding as ob
    mtype = ob.FIVE_STATE if cfg["model"] == "fiveState" else ob.THREE_STATE
    op = ob.params(diagonalExpansion=cfg["expansion"], splitMatrixBiggerThanThis=cfg.get("split", 10 ** 15))
    rates, cells, dts = [], 0, []
    rg = bool(cfg.get("ragged"))
    for _ in range(reps):
        t0 = time.time()
        if cfg.get("emit") == "expect":
            acc = ob.hmm(mtype, 0.0)
            cells = ob.batch_expectations(ob.model(mtype), problems, op, acc, rg, rg, threads=threads)
        else:
            _, cells = ob.batch_aligned_pairs(ob.model(mtype), problems, op, rg, rg, threads=threads)
        dts.append(time.time() - t0)
        rates.append(cells / dts[-1])
    rates.sort()
    return rates[len(rates) // 2], cells, sorted(dts)[len(dts) // 2]


def usable_cores():
    """Cores this process may actually use: the affinity mask, cut to the cgroup's CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) or os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def other_configs(names):
    """Short passes of the other BASELINE configs, one child process each (this process keeps the GPU context; the
    children are started, not exec'ed).  Per config: kernel-only cells/s, ms per step, roofline fraction, end-to-end cells/s."""
    res = {}
    for name in names:
        t0 = time.perf_counter()
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--e2e-batches", "25" if name == "4" else "5", "--no-other-configs"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
            d = json.loads(line)
            res[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
                         "frac": d["roofline"]["frac"], "value_e2e": d["value_e2e"], "pipeline_depth": d["e2e"].get("pipeline_depth"),
                         "workload": d["config"]["workload"], "wall_s": time.perf_counter() - t0}
        except Exception as e:  # noqa: BLE001 -- a failed side pass must not lose the headline line
            res[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="B", choices=["A", "B", "4", "5", "plumbing"])
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="default: strong when --gpus > 1 (the config's batch dealt out over the ranks)")
    ap.add_argument("--pairs", type=int, default=0, help="override the config's number of pairs (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--anchor-triples", action="store_true",
                    help="config 4: hand the anchors over as one (x, y, expansion) triple per column instead of as runs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (host to host) measurements")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default run (config B, one GPU) appends short passes of configs A, 4 and 5 as `other_configs`; skip them")
    ap.add_argument("--e2e-batches", type=int, default=0, help="batches pushed through the pipeline (odd: the steady state is read over an even number of batch intervals); 0: nine, or as many as make the section last ~0.5 s (short batches: one hiccup of a few ms must not decide the figure)")
    ap.add_argument("--e2e-depth", type=int, default=0, help="batches in flight in the end-to-end pipeline; 0: two, or three where the host's packing and planning takes at least half as long as sweep + download and three batches fit in half the device memory")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend: nccl is RCCL; gloo only for the CPU dry run of the launcher (tests)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: partition the batch, run the rendezvous and the reductions, print the line "
                         "(value null) -- checks the N > 1 launch path where there is no GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch
    from cpecan_amd import api, workload
    from cpecan_amd import dist as cdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.backend == "gloo" and not args.dry_run and torch.cuda.device_count() > 0:
        # a rehearsal of the multi-rank path on a box with fewer GPUs than ranks (gloo only: RCCL wants a GPU per rank)
        local_rank %= torch.cuda.device_count()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    dry = args.dry_run
    if not dry:
        if not torch.cuda.is_available() or api.device_count() < 1:
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and not dry:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    red_dev = torch.device("cuda", local_rank) if (not dry and (world == 1 or args.backend == "nccl")) else torch.device("cpu")

    cfg = dict(workload.CONFIGS[args.config])
    n_total = args.pairs or cfg["n_pairs"]
    # ---- which pairs this rank aligns: independent objects, no exchange on the data path ----
    if scaling == "strong":
        mine = cdist.lpt_assign(workload.pair_costs(args.config, n_total), world)[rank]  # the SAME batch over N ranks
    else:
        mine = np.arange(rank * n_total, (rank + 1) * n_total)                            # every rank its own batch
    expect = cfg.get("emit") == "expect"

    if dry:
        cells, kernel_ms, elapsed = int(workload.pair_costs(args.config, n_total)[mine % n_total].sum()), None, 1.0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c = torch.tensor([cells, len(mine)], dtype=torch.int64)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            total_cells, total_pairs = int(c[0]), int(c[1])
        else:
            total_cells, total_pairs = cells, len(mine)
        if rank == 0:
            print(json.dumps({"metric": "band DP cells/s (banded fwd+bwd+posterior)", "value": None, "unit": "cells/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": scaling,
                              "dry_run": True, "config": {"workload": "config %s" % args.config, "pairs_total": total_pairs,
                                                          "estimated_cells_total": total_cells,
                                                          "world_size": world, "backend": "gloo"}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    torch.zeros(1, device="cuda")  # create the HIP context now: a once-per-process cost, not part of any upload
    torch.cuda.synchronize()
    sm, params, mtype = model_and_params(api, cfg)
    emit = api.EMIT_EXPECT if expect else api.EMIT_MATCH
    problems = workload.config_problems(args.config, mine)
    # the C-ABI array a C caller would hold; the realign configuration holds its anchors as runs of matching columns --
    # what cPecanRealign.c:525-529 makes of a cigar's match operations -- and hands them over as such
    # (cpecan_batch_add_many_runs; --anchor-triples: one (x, y, expansion) triple per column, as round 3 measured)
    as_runs = bool(cfg.get("realign")) and not args.anchor_triples
    prepared, n_prepared, _keep = (api.Batch.prepare_problems_runs if as_runs else api.Batch.prepare_problems)(problems)

    def make_batch():
        b = api.Batch(sm, params, emit=emit, device=local_rank)
        b.add_prepared(prepared, n_prepared)
        b.upload()
        return b

    t0 = time.perf_counter()
    batch = make_batch()
    upload_s = time.perf_counter() - t0
    st = batch.stats()
    cells = st.cells
    # An explicit (non-blocking) stream, not the legacy null stream: work on the null stream synchronises implicitly with
    # other streams, which would serialise a batch's sweep with the copies of the batches before and after it.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        batch.run(stream.cuda_stream)
        if expect:
            # the E-step ends with the counts on the host and, across ranks, summed: cPecanEm.py:184-188
            batch.download()
            acc = api.hmm_constructEmpty(0.0, mtype)
            batch.expectations(acc)
            if world > 1:
                cdist.allreduce_hmm(acc, red_dev)

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kernel_ms_sum = 0.0
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
        if expect:
            kernel_ms_sum += batch.stats().kernelMs
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    # average launch duration of the sweep kernels from HIP events on the launch stream (the expectation step's host part
    # sits between its launches: there the library's own start/stop events around each run are summed instead)
    kernel_ms = (kernel_ms_sum if expect else ev0.elapsed_time(ev1)) / max(1, args.steps)

    total_cells, total_pairs = cells, len(mine)
    per_rank_cells = [cells]
    per_rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        tr = torch.zeros(world, dtype=torch.float64, device=red_dev)
        tr[rank] = elapsed / args.steps * 1e3
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(v) for v in tr.tolist()]  # every rank's own time per step (the line's time is their maximum)
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.zeros(world, dtype=torch.int64, device=red_dev)
        c[rank] = cells
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        per_rank_cells = [int(v) for v in c.tolist()]
        total_cells = sum(per_rank_cells)
        n = torch.tensor([len(mine)], dtype=torch.int64, device=red_dev)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        total_pairs = int(n.item())

    # ---- results back on the host once: PCIe + list assembly (not in `value`), and a parity spot check below ----
    t1 = time.perf_counter()
    batch.download()
    d2h_s = time.perf_counter() - t1
    st = batch.stats()

    # ---- SURVEY 8d's wall clock: problems resident in host memory -> result lists materialised in host memory ----
    e2e = {"plan_upload_s": upload_s, "download_and_assemble_s": d2h_s, "pairs_emitted": int(st.pairs),
           "device_bytes": int(st.deviceBytes), "waves": int(st.wavesPerLaunch),
           # which form the widest size class ran in: the split forms hold whole regions' forward values (cpecan_hip.h)
           "launch_form": {0: "one wave per region", 1: "two launches", 2: "one launch"}[int(st.launchForm) & 3] +
                          (", absolute positions" if int(st.launchForm) & 4 else "")}
    value_e2e = None
    if not args.no_e2e:
        # the timed batch goes first: a config-B batch in its one-launch form holds the forward values of every region
        # (68 GB at config B); the pipeline below gets that memory for its own batches
        spot = [] if expect else [batch.result(i) for i in range(min(4, len(problems)))]  # for the parity spot check below
        batch.close()
        batch = None
        barrier()
        # untimed: the library pins a host block when it is reused (second life), so the first batches of a process pay for
        # the pinning of their blocks once; a steady stream of batches does not
        for _ in range(3):
            wb0 = make_batch()
            wb0.run(stream.cuda_stream)
            wb1 = make_batch()
            wb1.run(stream.cuda_stream)
            wb0.download()
            wb0.close()
            wb1.download()
            wb1.close()
        barrier()
        # (a) one batch after the other: create + pack + plan + upload + run + gather + download, nothing overlapped
        t0 = time.perf_counter()
        b = make_batch()
        t1 = time.perf_counter()
        b.run(stream.cuda_stream)
        b.download()
        t2 = time.perf_counter()
        e2e["serial_s_per_batch"] = t2 - t0
        # the two host stages in the steady state (the figures above are the FIRST batch of the process: cold device
        # memory, unpinned host blocks)
        e2e["first_batch_plan_upload_s"] = e2e.pop("plan_upload_s")
        e2e["plan_upload_s"] = t1 - t0
        e2e["run_download_assemble_s"] = t2 - t1
        b.close()
        # (b) several batches in flight from ONE host thread: while the sweep of batch k runs, batch k+1 is packed, planned
        # and uploaded and batch k-1 is gathered and downloaded (every batch works on streams and events of its own; its
        # download runs on the helper thread the library gives every batch).  Two in flight are enough where the sweep is
        # the longest stage (config B); where the host's packing and planning is (config 4), the latency of one batch --
        # upload tail + sweep + gather + fetch -- exceeds the host's interval and a third batch in flight hides it.
        nb = args.e2e_batches
        if nb <= 0:
            nb = max(9, min(201, int(0.5 / max(e2e["serial_s_per_batch"], 1e-4)))) | 1
        nb = max(3, nb)
        depth = args.e2e_depth
        if depth < 2:
            # as many batches in flight as the latency of one batch (pack + plan + upload, copy in, sweep, gather, copy out:
            # the serial figure above) holds of its longest stage -- the host's share or the sweep -- between two and four,
            # and no more than fit in half the device's memory
            total_mem = torch.cuda.get_device_properties(local_rank).total_memory
            stage = max(e2e["plan_upload_s"], kernel_ms * 1e-3, 1e-4)
            depth = max(2, min(4, int(np.ceil(e2e["serial_s_per_batch"] / stage))))
            if e2e["plan_upload_s"] > kernel_ms * 1e-3:
                # host-bound (config 4): in the pipeline a batch's latency is longer than the serial figure -- its gather and
                # its copy out queue behind the next batches' sweeps and copies -- and two more batches in flight cover that
                # (steady 36-42 ms per batch at four, 32-34 at six: profiles/r04_config4_chain_bound.txt)
                depth = min(6, depth + 2)
            while depth > 2 and depth * e2e["device_bytes"] > total_mem // 2:
                depth -= 1
        e2e["pipeline_depth"] = depth
        trace = os.environ.get("CPECAN_BENCH_TRACE") == "1"
        if depth > 2:
            # untimed: `depth` batches alive at once need `depth` sets of host and device blocks; the sets beyond the two of
            # the warm-up above are allocated -- and pinned on their second use -- here, not inside the timed pipeline
            for _ in range(2):
                ws = []
                for _k in range(depth + 1):  # the pipeline holds `depth` batches and the one being packed
                    wb = make_batch()
                    wb.run(stream.cuda_stream)
                    wb.download_begin()          # ... and their result lists, all at once
                    ws.append(wb)
                for wb in ws:
                    wb.download_end()
                for wb in ws:
                    wb.close()
        barrier()
        t0 = time.perf_counter()
        inflight, pairs_out, t_first, done = [], 0, None, 0
        def retire():
            nonlocal pairs_out, t_first, done
            old = inflight.pop(0)
            old.download_end()
            t = time.perf_counter()
            if t_first is None:
                t_first = t  # the first batch's lists are on the host: the pipeline is full from here
            pst = old.stats()
            pairs_out += int(pst.pairs)
            old.close()
            done += 1
            if trace and rank == 0:
                print("pipeline: batch %d on the host at %.1f ms (its kernel %.1f ms, d2h %.1f ms)"
                      % (done, 1e3 * (t - t0), pst.kernelMs, pst.d2hMs), file=sys.stderr, flush=True)
            return t
        for _ in range(nb):
            ta = time.perf_counter()
            cur = make_batch()
            tb = time.perf_counter()
            cur.run(stream.cuda_stream)
            cur.download_begin()  # the batch's helper thread waits for the sweep, gathers and fetches (cpecan_batch_download_begin)
            tc = time.perf_counter()
            inflight.append(cur)
            if len(inflight) >= depth:
                retire()
            if trace and rank == 0:
                print("pipeline host: pack+plan+upload %.1f ms, launch %.1f, retire %.1f" % (
                    1e3 * (tb - ta), 1e3 * (tc - tb), 1e3 * (time.perf_counter() - tc)), file=sys.stderr, flush=True)
        while inflight:
            t_last = retire()
        pipe_s, steady_s = t_last - t0, (t_last - t_first) / (nb - 1)
        if world > 1:
            t = torch.tensor([pipe_s, steady_s], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pipe_s, steady_s = float(t[0]), float(t[1])
        e2e["pipelined_batches"] = nb
        e2e["pipelined_total_s_per_batch"] = pipe_s / nb    # fill and drain of the pipeline included
        e2e["pipelined_steady_s_per_batch"] = steady_s      # lists of batch k on the host -> lists of batch k+1 on the host
        value_e2e = total_cells / steady_s

    if rank == 0:
        S = 5 if cfg["model"] == "fiveState" else 3
        bytes_per_cell = 16 * S + 8  # F written once + read once (2*S*8 B) + one posterior word
        achieved = cells * bytes_per_cell / (kernel_ms * 1e-3) / 1e9
        traffic = committed_profile("traffic", args.config, len(mine))
        compute = committed_profile("compute", args.config, len(mine))
        kname = "cpecan_pairhmm_%s<%d, ...>" % ("packed" if cfg.get("realign") else "sweep", S)
        roofline = {
            # the contract's figures: ALGORITHMIC bytes per launch / average launch duration, against the HBM peak
            "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
            "algorithmic_bytes_per_cell": bytes_per_cell, "kernel_ms": kernel_ms,
            # what the counters say moves: far fewer bytes than the algorithmic figure, so HBM is NOT what limits it
            "measured_hbm_gbs": (traffic["hbm_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9) if traffic else None,
            "limiter": "per-wave dependency chains (four logAdds deep per cell, an LDS round trip in each) at the 2-2.5 waves "
                       "per SIMD the rolling rows in LDS allow, then the LDS array and fp64 VALU issue (DESIGN.md section 5); "
                       "HBM traffic is ~1/3 of the algorithmic bytes",
        }
        if compute:
            # vector-instruction ceiling: VALU instructions per 64-cell group (PMC) x issue cycles measured per class
            # (profiles/r02_valu_rate.txt) -> cells/s with every SIMD issuing back to back
            roofline["compute"] = {"ceiling_cells_per_s": compute["ceiling_cells_per_s"],
                                   "frac": (cells / (kernel_ms * 1e-3)) / compute["ceiling_cells_per_s"],
                                   "valu_cycles_per_64_cells": compute["valu_cycles_per_64_cells"],
                                   "source": compute.get("source")}
        out = {
            "metric": "band DP cells/s (banded fwd+bwd+posterior)",
            "value": total_cells * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            # the arithmetic the path computes in: fp64 throughout; the expectation emitter's events are 2^(x log2 e) with the
            # exponent handed to v_exp_f32 as an fp32 value (~1e-7 per event; the sums and the whole DP stay fp64)
            "dtype": "f64 DP, f32 event exponent" if expect else "f64",
            "data": "synthetic",
            "config": {
                "workload": "config %s: %d pairs%s, %s, %s, diagonalExpansion=%d, %s, threshold=0.01, traceback 1000/40%s"
                            % (args.config, n_total, " per GPU" if scaling == "weak" and world > 1 else " in all",
                               ("%d-%d bp" % (cfg["min_len"], cfg["max_len"])) if cfg.get("realign")
                               else "%d x ~%d bp" % (cfg["length"], cfg["length"]),
                               cfg["model"], cfg["expansion"],
                               "anchors on every matching column (given as %s), split at gaps of %d, ragged ends"
                               % ("runs" if as_runs else "per-column triples", cfg["split"])
                               if cfg.get("realign") else ("anchors every 50 bp" if cfg["anchors"] else "no anchors"),
                               ", expectation emitter%s" % (" + all-reduce of %d counts" % (S * S + S * 16 + 1) if world > 1 else "")
                               if expect else ""),
                "pairs_total": total_pairs,
                "cells_total": total_cells,
                "cells_per_rank": per_rank_cells,
                "ms_per_step_per_rank": per_rank_ms,
                "world_size": world,
                "parallelism": "pairs dealt longest-first over %d GPU(s), no data-path collective" % world
                               if scaling == "strong" else "every GPU its own batch, no data-path collective",
            },
            "roofline": roofline,
            "value_e2e": value_e2e,
            "e2e": e2e,
        }
        if not args.no_cpu_baseline and world == 1:
            # (rank 0, N=1 only) every core the box gives this process: its affinity mask cut to its cgroup CPU quota (a
            # one-GPU box hands out a 16-core share of one socket; BASELINE.md section 3 asks for a whole socket's cores)
            threads = int(os.environ.get("CPECAN_BENCH_CPU_THREADS", usable_cores()))
            per_thread = 64 if not cfg.get("realign") else 256
            sample = problems[:min(len(problems), per_thread * threads)]
            v, ccells, dt = cpu_baseline(cfg, sample, threads)
            out["cpu_baseline"] = {
                "value": v, "unit": "cells/s", "cores": threads, "kind": "port", "cpu": cpu_model_name(),
                "sample": "first %d pairs of the same workload (%d cells), the oracle's C restatement with OpenMP over pairs "
                          "on %d threads (every core this process may use: affinity cut to the cgroup quota), median of 3 runs of %.1f s; the oracle does "
                          "~6.8e6 cells/s/core where the survey measured 4.6e6 for the reference itself: it flatters the CPU"
                          % (len(sample), ccells, threads, dt),
            }
            # parity spot check of the timed batch against the oracle (first 4 pairs)
            import oracle_binding as ob
            from parity import assert_pairs_match
            om = ob.model(ob.FIVE_STATE if cfg["model"] == "fiveState" else ob.THREE_STATE)
            op = ob.params(diagonalExpansion=cfg["expansion"], splitMatrixBiggerThanThis=cfg.get("split", 10 ** 15))
            if not expect:
                for i in range(min(4, len(problems))):
                    sx, sy, a, rl, rr = problems[i]
                    want = ob.aligned_pairs(om, sx, sy, a, op, rl, rr)
                    assert_pairs_match(spot[i] if batch is None else batch.result(i), want, threshold=params.threshold)
                out["parity_spot_check"] = "4 pairs match the oracle"
            else:
                # the expectation emitter: a batch of the first 32 problems against the oracle's counts (1e-5 relative)
                nchk = min(32, len(problems))
                acc = api.hmm_constructEmpty(0.0, mtype)
                with api.Batch(sm, params, emit=api.EMIT_EXPECT, device=local_rank) as cb:
                    cb.add_many(problems[:nchk])
                    cb.upload()
                    cb.run()
                    cb.download()
                    cb.expectations(acc)
                oacc = ob.hmm(ob.FIVE_STATE if cfg["model"] == "fiveState" else ob.THREE_STATE, 0.0)
                rgd = bool(cfg.get("ragged"))
                ob.batch_expectations(om, problems[:nchk], op, oacc, rgd, rgd, threads=threads)
                for i in range(S * S):
                    assert abs(acc.transitions[i] - oacc.T[i]) <= 1e-5 * abs(oacc.T[i]) + 1e-9, ("T", i)
                for i in range(S * 16):
                    assert abs(acc.emissions[i] - oacc.E[i]) <= 1e-5 * abs(oacc.E[i]) + 1e-9, ("E", i)
                assert abs(acc.likelihood - oacc.likelihood) <= 1e-9 * abs(oacc.likelihood)
                out["parity_spot_check"] = "expectation counts of %d problems match the oracle to 1e-5" % nchk
        if (args.config == "B" and world == 1 and not args.pairs and not args.no_other_configs and not args.no_e2e):
            # the driver's default run sees the other BASELINE configs too -- short passes (3 steps, a
            # five-batch pipeline) as child processes once this process has given its device memory back
            if batch is not None:
                batch.close()
                batch = None
            api.cache_trim(local_rank)  # (not -1: that would touch -- and create a context on -- every GPU of the node)
            out["other_configs"] = other_configs(("A", "4", "5"))
        print(json.dumps(out), flush=True)
    if batch is not None:
        batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

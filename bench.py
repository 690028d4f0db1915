#!/usr/bin/env python3
"""bench.py -- bundles/sec of the MI355X splice-graph decomposition path (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic splice graphs that is already resident in HBM:
kernel launch(es) on the batch's HIP stream, D2H of the status words + packed path records and, at N > 1, the
RCCL gather of the records to rank 0 (the path's only exchange step: SURVEY.md 8e).

Workload (N=1): BASELINE.json configs[1] -- 100k synthetic splice graphs, 64 vertices / 256 edges each.
N > 1: bundles shard embarrassingly; every rank decomposes its own 100k-graph shard (weak scaling, seed 1004+rank).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--graphs G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes per launch (packed input +
packed path records) / mean kernel duration measured with HIP events on the launch stream.  `cpu_baseline` = the
oracle (CPU restatement of the reference, oracle/) on a bounded sample of the same workload, all host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(pg_sample, cores: int):
    """The oracle ("port": our CPU restatement of the reference algorithm) on the GPU box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    t0 = time.time()
    _, _, sec, _ = common.oracle_run(pg_sample, threads=cores)
    wall = time.time() - t0
    return pg_sample.n / sec, sec, wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--graphs", type=int, default=100000, help="graphs per GPU (BASELINE configs[1]: 100k)")
    ap.add_argument("--vertices", type=int, default=64)
    ap.add_argument("--edges", type=int, default=256)
    ap.add_argument("--weights", choices=("uniform", "int", "flow"), default="uniform",
                    help="edge weights: uniform = U[1,100) f64 (BASELINE configs), flow = flow-conserving sums of s-t paths (SURVEY.md 8d's second distribution)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="graphs in the bounded cpu_baseline sample (0 = skip)")
    args = ap.parse_args()
    # stdout carries exactly one line, the JSON: libraries that write banners to file descriptor 1 (RCCL prints its version block
    # there at init) are sent to stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1); os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    dist_on = world > 1 or bool(os.environ.get("ALD_BENCH_FORCE_DIST"))      # the override runs the exchange path with one rank (rehearsal on a 1-GPU box)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    import aletsch_amd as A

    dev = local if dist_on else 0
    seed = 1002 if world == 1 else 1004 + rank          # SURVEY.md 8d seeds: cfg2 = 1002, cfg4 = 1004 + rank
    pg = A.synth(seed=seed, n_graphs=args.graphs, v_min=args.vertices, v_max=args.vertices, fixed_edges=args.edges,
                 weight_mode={"uniform": 0, "int": 1, "flow": 2}[args.weights])
    # Two batch objects over the same resident input, used alternately: the D2H of batch k's path records (SDMA, own stream)
    # overlaps batch k+1's kernel.  Kernels never overlap each other (the previous one is synchronised before the next launch),
    # so the per-launch HIP-event time stays the time of ONE kernel on an otherwise idle GPU.
    batches = [A.DecompBatch(dev), A.DecompBatch(dev)]
    for b in batches:
        b.add(pg)
        b.upload()                                      # inputs resident in HBM before the timed region
    batch = batches[0]

    from aletsch_amd.distributed import RecordGatherer, _device_words
    gatherer = RecordGatherer(torch.device("cuda", dev)) if dist_on else None

    pool_read = {}                                       # batch -> event reached once the exchange has read its record pool

    def gather_records(b):
        """RCCL gather of the packed path records to rank 0, straight from the batch's record pool in HBM (no host round trip);
        only enqueued: the next kernel is launched behind it and the host never waits for a collective"""
        ptr, n = b.device_records()
        pool_read[id(b)] = gatherer.gather(_device_words(ptr, n, torch.device("cuda", dev)), graph_offset=rank * args.graphs)

    def finish(b):
        b.download()                                    # stream sync + D2H of status / packed records (+ class retries)
        if dist_on:
            gather_records(b)
        return b.kernel_ms()

    def run_steps(k):
        """k complete passes (kernel + results on the host); returns the per-launch kernel times"""
        ms = []; prev = None
        for i in range(k):
            cur = batches[i % 2]
            if prev is not None:
                prev.sync()                             # the previous kernel is done ...
            ev = pool_read.pop(id(cur), None)
            if ev is not None:
                ev.synchronize()                        # (the exchange of this batch's previous records has read its pool)
            cur.run()                                   # ... the next one starts ...
            if prev is not None:
                ms.append(finish(prev))                 # ... while the previous records travel to the host
            prev = cur
        if prev is not None:
            ms.append(finish(prev))
        return ms

    run_steps(args.warmup)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = run_steps(args.steps)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        te = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    res = batch.result()
    n_bad = int((res.status != 0).sum())
    in_b, out_b = batch.algorithmic_bytes()
    k_ms = float(np.mean(kms))
    info = batch.class_info(1)

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process, so the figure comes from the committed
    # summary of separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command (profiles/summarize.py);
    # it is only quoted when it was collected on this exact workload, else null
    traffic = None; traffic_src = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tj.get("graphs_per_gpu") == args.graphs and tj.get("vertices") == args.vertices and tj.get("edges") == args.edges and args.weights == "uniform":
            traffic = tj["traffic_bytes_per_launch"]; traffic_src = tj.get("source")
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0:
        total_graphs = args.graphs * world * args.steps
        value = total_graphs / elapsed
        achieved = (in_b + out_b) / (k_ms / 1e3) / 1e9
        line = {
            "metric": "bundles/sec", "value": value, "unit": "bundles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.graphs} synthetic splice graphs per GPU, {args.vertices} vertices / {args.edges} edges each "
                                   + ("(BASELINE.json configs[1]; U[1,100) FP64 weights, 1 supporting sample per edge, no phasing paths)" if args.weights == "uniform" else
                                      f"(BASELINE.json configs[1] shape with {args.weights} weights -- secondary distribution, not the headline)"),
                       "graphs_per_gpu": args.graphs, "sharding": "independent graphs per rank, RCCL gather of path records to rank 0" if world > 1 else "single GPU",
                       "failed_graphs": n_bad, "paths_per_graph": float(len(res.weight)) / max(1, args.graphs),
                       "workgroups_per_cu": info["blocks_per_cu"], "grid": info["blocks_last_run"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_gbs": (traffic / (k_ms / 1e3) / 1e9) if traffic else None,      # measured HBM traffic over this run's kernel time
                         "kernel": "ald_decomp_kernel_c1", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": in_b + out_b,
                         "bytes_per_graph": (in_b + out_b) / args.graphs, "kernel_graphs_per_s": args.graphs / (k_ms / 1e3)},
        }
        if args.cpu_sample > 0 and world == 1:           # reported at N = 1 only
            # the GPU box gives each GPU a share of the host (16 cores per GPU): use the cores this process may actually run on
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(cores, 16 * max(1, world)))
            sample = pg.select(np.arange(min(args.cpu_sample, pg.n)))
            v, cpu_sec, wall = cpu_baseline(sample, cores)
            line["cpu_baseline"] = {"value": v, "unit": "bundles/s", "cores": cores, "kind": "port",
                                    "sample": f"first {sample.n} graphs of the same workload, oracle/ (CPU restatement of the reference scallop core), "
                                              f"{cores} threads over independent graphs, {cpu_sec:.2f} s"}
        sys.stdout.flush(); os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    for b in batches:
        b.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- bundles/sec of the MI355X splice-graph decomposition path (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic splice graphs, H2D + kernel + D2H inclusive: the
caller's host arrays go into the batch's pinned host arrays (ald_batch_add_packed) and over PCIe into HBM, the decomposition kernels
run (they join the exons of every path into its record and write the result index), the status words, the records -- paths AND
transcripts -- and the index come back and are decoded into the host's path table with coverage = log(1 + weight): graphs fully
decomposed, paths + transcripts materialised in host memory.  The stages are pipelined on three host threads over four rotating batch
objects, as a caller feeding a stream of batches would run them: when the timed region starts the batches of the first steps are
already resident in HBM (staged during the warm-up) and the staging of later ones runs under the kernels.  At N > 1 the step also
holds the RCCL gather of the finished transcripts to rank 0, the path's only exchange step.
`value` is that rate.  `value_resident` is the rate of a second loop over inputs that stay resident in HBM (kernels + D2H + decode
only, nothing staged inside the window); kernels never overlap each other, so the per-launch HIP-event time is that of one kernel,
and `roofline` is computed from the launches of the resident loop.

Workload (N=1): BASELINE.json configs[1] -- 100k synthetic splice graphs, 64 vertices / 256 edges each.
N > 1: bundles shard embarrassingly; every rank decomposes its own 100k-graph shard (weak scaling, seed 1004+rank).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--graphs G]
        (--gpus N > 1 without a launcher's WORLD_SIZE in the environment: this process starts the N ranks itself, as
         `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...`, before it touches the GPU,
         relays rank 0's JSON line and exits with the launcher's code)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes per launch (packed input + packed path
records) / mean kernel duration measured with HIP events on the launch stream over the timed region.  `cpu_baseline` = the oracle
(CPU restatement of the reference, oracle/) on a bounded sample of the same workload on the host cores, 1 thread and all cores.
"""
from __future__ import annotations

import argparse
import json
import os
import queue
import socket
import subprocess
import sys
import threading
import time

# A batch owns seven HIP streams (one for copies / ordering, six that its size classes are dealt to) and a pipelined caller keeps
# several batches in flight; the ROCm runtime maps all streams of a process onto FOUR hardware queues by default, so the D2H copy of
# batch k regularly sat in the same queue as the kernel of batch k+1 and waited for it (download 24-38 ms instead of 3 ms per step).
# Must be in the environment before the HIP runtime initialises, i.e. before anything touches the GPU; ranks inherit it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_CUS = 256


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graphs", type=int, default=100000, help="graphs per GPU (BASELINE configs[1]: 100k)")
    ap.add_argument("--vertices", type=int, default=64)
    ap.add_argument("--edges", type=int, default=256)
    ap.add_argument("--weights", choices=("uniform", "int", "flow"), default="uniform",
                    help="edge weights: uniform = U[1,100) f64 (BASELINE configs), flow = flow-conserving sums of s-t paths (SURVEY.md 8d's second distribution)")
    ap.add_argument("--cpu-sample", type=int, default=32768, help="graphs in the bounded all-cores cpu_baseline sample (0 = skip both CPU legs)")
    ap.add_argument("--skip-h2d-loop", action="store_true", help="profiling runs: make the batches resident with one plain pass each and skip the host-arrays-in loop (`value` is then the resident-input rate and the line says so)")
    ap.add_argument("--serial-steps", action="store_true", help="profiling runs: download every batch before the next kernel starts (under rocprofv3 the D2H copies run as blit kernels on the CUs and would share them with the decomposition kernel)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg3 / flow-weight kernel timings (the `secondary` block)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="gloo: CPU rehearsal of the multi-rank plumbing (needs --dry-run)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: every rank stages its shard and enters the exchange with an empty stream (CPU tier)")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """--gpus N without a launcher: start the N ranks as fresh child processes (this process has not touched torch or HIP yet and
    never will), relay rank 0's JSON line, propagate the launcher's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    if r.returncode == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no result line\n")
        return 1
    return r.returncode


def source_stamp() -> dict:
    """Which source this line was measured on: the git HEAD (read from .git here, or from build/git_head.txt -- written by
    tools/stamp_head.sh before a gpurun call, whose snapshot carries no .git), and content hashes of the kernel source and of the
    loaded product library.  profiles/summarize.py copies the block into every summary it writes."""
    import hashlib
    def sha16(paths):
        h = hashlib.sha256()
        for q in paths:
            try:
                h.update(open(os.path.join(ROOT, q), "rb").read())
            except OSError:
                h.update(b"missing:" + q.encode())
        return h.hexdigest()[:16]
    head = None
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10)
        if r.returncode == 0:
            d = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10)
            head = r.stdout.strip()[:12] + ("+dirty" if d.stdout.strip() else "")
    except (OSError, subprocess.SubprocessError):
        pass
    if head is None:
        try:
            head = open(os.path.join(ROOT, "build", "git_head.txt")).read().strip()
        except OSError:
            head = "unknown"
    return {"git_head": head,
            "kernel_source_sha16": sha16(["aletsch_amd/csrc/decomp_device.h", "aletsch_amd/csrc/decomp_common.h", "aletsch_amd/csrc/decomp_class.hip"]),
            "library_sha16": sha16(["aletsch_amd/lib/libaletsch_decomp.so"]), "bench_sha16": sha16(["bench.py"])}


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_legs(pg, n_all: int):
    """The oracle ("port": our CPU restatement of the reference algorithm) on the GPU box's host cores: one thread, and all the
    cores this process may run on, over bounded samples of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    cores, affinity, quota = cpu_cores(detail=True)
    n_all = min(n_all, pg.n); n_one = max(256, min(pg.n, n_all // 12))
    s_all = pg.select(np.arange(n_all)); s_one = pg.select(np.arange(n_one))
    _, _, sec_all, _ = common.oracle_run(s_all, threads=cores)
    _, _, sec_one, _ = common.oracle_run(s_one, threads=1)
    model = cpu_model()
    note = ("oracle/ = CPU restatement of the reference scallop core (container-based port; the survey's probe of the reference itself "
            "measured 157-206 graphs/s per thread on these graphs, BASELINE.md section 2)")
    allc = {"value": s_all.n / sec_all, "unit": "bundles/s", "cores": cores, "kind": "port", "cpu_model": model,
            "affinity_cores": affinity, "cgroup_cpu_quota": quota,
            "sample": f"first {s_all.n} graphs of the same workload, {cores} threads over independent graphs (affinity mask {affinity} hardware threads, cgroup CPU quota {quota}), {sec_all:.2f} s; {note}"}
    one = {"value": s_one.n / sec_one, "unit": "bundles/s", "cores": 1, "kind": "port", "cpu_model": model,
           "sample": f"first {s_one.n} graphs of the same workload, 1 thread, {sec_one:.2f} s"}
    return allc, one


def cpu_cores(detail=False):
    """every core this process may USE (see below): -> cores, or (cores, affinity, quota)"""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    # every core this process may USE: the affinity mask, cut down to the cgroup's CPU quota where there is one (the GPU box shows all
    # 256 hardware threads in the mask but gives a one-GPU job a 16-core share; 256 threads on 16 cores' worth of time ran the same
    # sample at 5.9 k bundles/s against 10.4 k with 16)
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1]))),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (None if int(t) <= 0 else int(t) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))):
        try:
            quota = parse(open(path).read().strip())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    cores = max(1, min(affinity, int(quota + 0.5)) if quota else affinity)
    if os.environ.get("ALD_BENCH_CPU_THREADS"):
        cores = max(1, int(os.environ["ALD_BENCH_CPU_THREADS"]))
    return (cores, affinity, quota) if detail else cores


def committed_pmc(args):
    """HBM traffic and instruction counts of the dominant kernel: PMC counters cannot be read from inside the process, so the figures
    come from the committed summary of separate `rocprofv3 --pmc` passes over this same command (profiles/summarize.py); they are only
    quoted when they were collected on this exact workload, else null."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tj.get("graphs_per_gpu") == args.graphs and tj.get("vertices") == args.vertices and tj.get("edges") == args.edges and args.weights == "uniform":
            return tj
    except (OSError, ValueError, KeyError):
        pass
    return {}


def kernel_only(A, dev, pg, reps=4):
    """upload once, run `reps` times, -> (best kernel ms, failed graphs, per-class graph counts, algorithmic bytes in + out); four runs: the
    first run of a batch object also sizes its slabs, and single runs of the mixed batch spread over 88-92 ms"""
    with A.DecompBatch(dev) as b:
        b.add(pg); b.upload()
        best = None
        for _ in range(reps):
            b.run(); b.download()
            ms = b.kernel_ms(); best = ms if best is None else min(best, ms)
        bad = int((b.result().status != 0).sum())
        classes = {str(c): b.class_info(c)["n_graphs"] for c in range(14) if b.class_info(c)["n_graphs"]}
        in_b, out_b = b.algorithmic_bytes()
    return best, bad, classes, in_b + out_b


def secondary_roofline(alg_bytes, ms, n_graphs, kernels):
    """the same roofline object as the headline's, for a secondary shape: algorithmic bytes of the batch (ald_batch_algorithmic_bytes:
    packed input + packed path records) over the HIP-event time from the first launch to the last kernel's end"""
    achieved = alg_bytes / (ms / 1e3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": kernels, "kernel_ms": ms, "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_graph": alg_bytes / max(1, n_graphs)}


def secondary_cpu_baseline(pg, cores_hint, seconds=8.0, probe=256):
    """the oracle ("port") on a bounded sample of a secondary shape: a probe of `probe` graphs sizes the sample to about `seconds` of
    all-core work (graphs are taken with a stride over the whole batch, so the sample has the batch's size mix)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    cores = max(1, cores_hint)
    idx = np.linspace(0, pg.n - 1, num=min(pg.n, probe)).astype(np.int64)
    _, _, sec_p, _ = common.oracle_run(pg.select(np.unique(idx)), threads=cores)
    n = int(min(pg.n, max(probe, len(np.unique(idx)) * seconds / max(sec_p, 1e-3))))
    idx = np.unique(np.linspace(0, pg.n - 1, num=n).astype(np.int64))
    s = pg.select(idx)
    _, _, sec, _ = common.oracle_run(s, threads=cores)
    return {"value": s.n / sec, "unit": "bundles/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{s.n} graphs taken with an even stride over the batch (same size mix), {cores} threads over independent graphs, {sec:.2f} s"}


def sink_pipeline(A, pg, n, rounds=6):
    """host arrays in -> merged transcript set out, stages overlapped on four host threads (stage | kernel | download | merge into ONE
    persistent set that keeps growing): the rate an integrator's whole loop would see, for both merge paths"""
    import numpy as np
    out = {}
    sid = (np.arange(n) % 8).astype(np.int32)
    for mode in ("host_sink", "gpu_reduction"):
        batches = [A.DecompBatch(0) for _ in range(4)]
        for b in batches:
            b.add(pg); b.upload(); b.run(); b.download(); b.clear()
        free = queue.Queue(); staged = queue.Queue(maxsize=1); ran = queue.Queue(maxsize=1); done = queue.Queue(maxsize=1)
        for b in batches:
            free.put(b)
        sink = A.TranscriptSink(0.8); err = []

        def stage():
            try:
                for _ in range(rounds):
                    b = free.get(); b.clear(); b.add(pg); b.upload(); staged.put(b)
            except BaseException as e:
                err.append(e)
            staged.put(None)

        def kern():                                           # the kernel of batch k + 1 runs while batch k is copied back and decoded
            try:
                while True:
                    b = staged.get()
                    if b is None:
                        break
                    b.run(); b.sync(); ran.put(b)
            except BaseException as e:
                err.append(e)
            ran.put(None)

        def fetch():
            try:
                while True:
                    b = ran.get()
                    if b is None:
                        break
                    b.download(); done.put(b)
            except BaseException as e:
                err.append(e)
            done.put(None)

        def merge():
            r = 0
            try:
                while True:
                    b = done.get()
                    if b is None:
                        return
                    if mode == "gpu_reduction":
                        b.reduce_into(sink, sid, tid_base=r << 44, skip_single_exon=True)
                    else:
                        sink.add_batch(b, sid, tid_base=r << 44, skip_single_exon=True)
                    r += 1; free.put(b)
            except BaseException as e:
                err.append(e)
                while done.get() is not None:
                    pass
        ths = [threading.Thread(target=f, daemon=True) for f in (stage, kern, fetch, merge)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        el = time.perf_counter() - t0
        if err:
            raise err[0]
        out[mode] = {"bundles_per_s": rounds * n / el, "ms_per_batch": 1e3 * el / rounds}
        sink.close()
        for b in batches:
            b.close()
    out["workload"] = f"{rounds} batches of {n} graphs through stage | kernel | download | merge into one persistent transcript set (skip_single_exon as the reference's default), four host threads"
    return out


def dispatcher_rate(submitters=8):
    """The call site an integrator would write: `submitters` pool threads hand reference-shaped objects (heap edges, std::set /
    unordered_map members per edge) to aletsch::gpu_assembly_queue, which converts, batches, runs and merges them into one transcript
    set (tools/dispatch_bench.cc, a C++ program over the C ABI; built here with g++ and run as a child process)."""
    import re, shutil
    exe = os.path.join(ROOT, "build", "dispatch_bench")
    lib = os.path.join(ROOT, "aletsch_amd", "lib")
    try:
        if shutil.which("g++") is None:
            return {"skipped": "no g++ on this box"}
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["g++", "-std=c++11", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "dispatch_bench.cc"), "-o", exe,
                        "-L" + lib, "-laletsch_decomp", "-Wl,-rpath," + lib], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        r = subprocess.run([exe, "20000", "50", str(submitters), "32768", "4", "1", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        m = re.search(r"measured: (\d+) graphs.*-> (\d+) graphs/s \(failed (\d+)", r.stdout)
        if r.returncode != 0 or not m:
            return {"skipped": "dispatch_bench failed", "rc": r.returncode, "tail": r.stdout[-300:]}
        return {"workload": f"{m.group(1)} graphs of the headline shape as reference-shaped objects, {submitters} submitting threads -> gpu_assembly_queue (batches of 32768, 4 slots, 2 pack threads) -> one transcript set",
                "bundles_per_s": float(m.group(2)), "failed_graphs": int(m.group(3)), "submitters": submitters}
    except Exception as e:                                     # a diagnostic leg must not take the bench line down
        return {"skipped": f"{type(e).__name__}: {e}"}


def main() -> int:
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        return launch_ranks(args)                        # before any torch / HIP call in this process
    rank = int(os.environ.get("RANK", "0")); world = int(world_env or "1"); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        return 2
    if args.dry_run != (args.backend == "gloo"):
        sys.stderr.write("bench.py: --backend gloo and --dry-run go together (the CPU rehearsal of the multi-rank plumbing)\n")
        return 2
    # stdout carries exactly one line, the JSON: libraries that write banners to file descriptor 1 (RCCL prints its version block
    # there at init) are sent to stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1); os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    dist_on = world > 1 or bool(os.environ.get("ALD_BENCH_FORCE_DIST"))      # the override runs the exchange path with one rank (rehearsal on a 1-GPU box)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
    import aletsch_amd as A
    from aletsch_amd.distributed import StreamGatherer, _device_words

    dev = local if dist_on else 0
    tdev = torch.device("cpu") if args.dry_run else torch.device("cuda", dev)
    seed = 1002 if world == 1 else 1004 + rank          # SURVEY.md 8d seeds: cfg2 = 1002, cfg4 = 1004 + rank
    pg = A.synth(seed=seed, n_graphs=args.graphs, v_min=args.vertices, v_max=args.vertices, fixed_edges=args.edges,
                 weight_mode={"uniform": 0, "int": 1, "flow": 2}[args.weights])
    gatherer = StreamGatherer(tdev) if dist_on else None

    def emit(line):
        sys.stdout.flush(); os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)

    if args.dry_run:
        # CPU rehearsal: no kernel exists here, so every rank stages its shard (host code of the product) and goes through the same
        # exchange step with a stream that only says who it is; rank 0 reports the world the collective actually saw
        me = np.array([rank, pg.n, int(pg.g_ne.sum())], np.int32)
        gatherer.gather(torch.from_numpy(me), graph_offset=rank * args.graphs)
        if rank == 0:
            st = gatherer.streams()
            seen = sorted(int(w.view(np.int32)[0]) for w, _ in st)
            emit({"metric": "bundles/sec", "value": None, "unit": "bundles/s", "n_gpus": len(st), "dry_run": True, "backend": "gloo",
                  "ranks_in_gather": seen, "graph_offsets": [int(o) for _, o in st], "graphs_staged": [int(w.view(np.int32)[1]) for w, _ in st]})
        dist.barrier(); dist.destroy_process_group()
        return 0

    NB = 6 if dist_on else 4                             # in flight: one kernel, one download, two being staged (+ two in the exchange: its kernels wait for room beside the decomposition kernel)
    batches = [A.DecompBatch(dev) for _ in range(NB)]

    xs = [0.0, 0.0, 0.0, 0]                              # exchange thread: stream built on the device / gather enqueued / stream read by the gather

    def exchange(b):
        """RCCL gather of this batch's finished transcripts to rank 0.  The stream (final records only, exons joined) is built by
        kernels and stays in HBM (ald_batch_device_transcript_stream); a zero-copy view of it goes into the gather, so the transcripts
        travel HBM -> xGMI -> HBM of rank 0.  Runs on its own host thread (a fourth pipeline stage): the thread that launches the
        kernels never waits for a collective, and the batch goes back to the stagers once the gather has read it."""
        t0_ = time.perf_counter()
        ptr, n = b.device_transcript_stream()
        t1_ = time.perf_counter()
        t = _device_words(ptr, n, tdev) if n else torch.zeros(0, dtype=torch.int32, device=tdev)
        read = gatherer.gather(t, graph_offset=rank * args.graphs)
        t2_ = time.perf_counter()
        if read is not None:
            read.synchronize()
        xs[0] += t1_ - t0_; xs[1] += t2_ - t1_; xs[2] += time.perf_counter() - t2_; xs[3] += 1

    dl_acc = {"wait_kernel": 0.0, "status_retries": 0.0, "copy": 0.0, "decode": 0.0, "bytes_to_host": 0, "n": 0}

    def finish(b):
        b.download()                                    # stream sync + D2H of status / records (paths + transcripts) / index (+ class retries) + decode
        d = b.download_ms()
        for k_ in ("wait_kernel", "status_retries", "copy", "decode", "bytes_to_host"):
            dl_acc[k_] += d[k_]
        dl_acc["n"] += 1
        return b.kernel_ms()

    def bracket():
        """the contract's bracket: everything issued so far is done on every rank"""
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        return time.perf_counter()

    def run_steps(w, k, staged, raw=False, prime=False):
        """w untimed + k timed passes through ONE pipeline.  staged=True: host arrays in -> host results out, three stages on three
        threads (adder: caller's arrays -> the batch's host arrays; uploader: H2D + first-pass work lists; this thread: kernel
        launch + D2H), batch objects rotating; staged=False: the batches are resident, kernel + D2H only.
        The timed window is a window over the pipeline's steady state: between the two brackets (GPU idle, all results of the
        previous step on the host, barrier across ranks) exactly k kernels run, k results are downloaded and k batches are staged --
        the stagers, like a caller feeding a stream of batches, work ahead by up to two batches, so the first timed kernels consume
        batches staged during the warm-up and the last two stagings inside the window are for batches beyond it (staged, never run).
        Returns (seconds between the brackets, per-launch kernel ms of the timed steps)."""
        ahead = 2 if staged else 0
        ready = queue.Queue(); free = queue.Queue()
        for b in batches:
            free.put(b)
        err = []
        todo = queue.Queue()
        for _ in range(w + k + ahead):
            todo.put(1)
        added = queue.Queue(); xq = queue.Queue()
        stage_s = [0.0, 0.0, 0.0, 0.0]                           # busy seconds: host copy / H2D / main thread waiting for a staged batch / exchange

        def exchanger():                                         # stage 4 (multi-GPU only): finished transcripts -> rank 0
            if tdev.type == "cuda":
                torch.cuda.set_device(tdev)
            while True:
                b = xq.get()
                try:
                    if b is None:
                        return
                    t_ = time.perf_counter()
                    exchange(b)
                    stage_s[3] += time.perf_counter() - t_
                    free.put(b)
                except BaseException as e:
                    err.append(e); ready.put(None)
                finally:
                    xq.task_done()

        def release(b):                                          # a downloaded batch: through the exchange when there is one, else straight back
            if dist_on:
                xq.put(b)
            else:
                free.put(b)

        def drained():                                           # every exchange issued so far has been enqueued (and has read its stream)
            xq.join()
            if err:
                raise err[0]

        def adder():                                             # stage 1: caller's arrays -> the batch's canonical host arrays (+ in-CSR)
            try:
                while True:
                    try:
                        todo.get_nowait()
                    except queue.Empty:
                        added.put(None); return
                    b = free.get()
                    t_ = time.perf_counter()
                    if staged:
                        b.clear()
                        if raw:
                            b.add_packed_raw(pg, 10000)      # the graphs as assemble(gx, px, sid) receives them: the pre-steps run in the kernel
                        else:
                            b.add(pg)
                    stage_s[0] += time.perf_counter() - t_
                    added.put(b)
            except BaseException as e:                           # surface the failure in the main thread instead of a hang
                err.append(e); added.put(None); ready.put(None)

        def uploader():                                          # stage 2: H2D of the batch arrays + first-pass work lists
            try:
                while True:
                    b = added.get()
                    if b is None:
                        return
                    t_ = time.perf_counter()
                    if staged:
                        b.upload()
                    stage_s[1] += time.perf_counter() - t_
                    ready.put(b)
            except BaseException as e:
                err.append(e); ready.put(None)
        ths = [threading.Thread(target=adder, daemon=True), threading.Thread(target=uploader, daemon=True)] + ([threading.Thread(target=exchanger, daemon=True)] if dist_on else [])
        for th in ths:
            th.start()
        prev = None; ms = []
        main_s = [0.0, 0.0, 0.0]                                 # main thread: waiting for the previous kernel / launching / download
        t0 = bracket() if w == 0 else None
        for step in range(w + k):
            t_ = time.perf_counter()
            cur = ready.get()
            stage_s[2] += time.perf_counter() - t_
            if cur is None:
                raise err[0]
            t_ = time.perf_counter()
            if prev is not None:
                prev.sync()                             # the previous kernel is done ...
            t1_ = time.perf_counter(); main_s[0] += t1_ - t_
            if args.serial_steps and prev is not None:
                x = finish(prev); release(prev); prev = None
                if step > w:
                    ms.append(x)
            cur.run()                                   # ... the next one starts ...
            t2_ = time.perf_counter(); main_s[1] += t2_ - t1_
            if prev is not None:
                x = finish(prev); release(prev)         # ... while the previous records travel to the host
                if step > w:
                    ms.append(x)
            main_s[2] += time.perf_counter() - t2_
            prev = cur
            if step == w - 1:                           # last warm-up step: drain it, then the opening bracket
                finish(prev); release(prev); prev = None
                drained()
                t0 = bracket(); stage_s[2] = 0.0; stage_s[3] = 0.0; main_s = [0.0, 0.0, 0.0]
        ms.append(finish(prev)); release(prev)
        drained()
        t1 = bracket()
        xq.put(None)
        for th in ths:
            th.join()
        el = t1 - t0
        if dist_on:
            te = torch.tensor([el], dtype=torch.float64, device=tdev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        if os.environ.get("ALD_BENCH_STAGES"):
            n_st = max(1, w + k + ahead)
            sys.stderr.write("[bench] %d + %d %s steps: add %.1f ms, upload %.1f ms per batch; timed steps: main waited for a staged batch %.1f ms; main thread: sync %.1f ms, launch %.2f ms, download %.1f ms; exchange thread %.1f ms (per step); kernel %.2f ms per launch (HIP events)\n"
                             % (w, k, "staged" if staged else "resident", 1e3 * stage_s[0] / n_st, 1e3 * stage_s[1] / n_st, 1e3 * stage_s[2] / k, 1e3 * main_s[0] / k, 1e3 * main_s[1] / k, 1e3 * main_s[2] / k, 1e3 * stage_s[3] / k, sum(ms) / max(1, len(ms))))
        if os.environ.get("ALD_BENCH_STAGES") and dist_on and xs[3]:
            sys.stderr.write("[bench]   exchange thread per batch: device stream %.1f ms, gather enqueue %.1f ms, wait until read %.1f ms\n" % (1e3 * xs[0] / xs[3], 1e3 * xs[1] / xs[3], 1e3 * xs[2] / xs[3]))
        return el, ms

    # one plain pass per batch object first (untimed, in front of the W warm-up steps): every pinned / device buffer exists, every batch is resident
    for b in batches:
        b.clear(); b.add(pg); b.upload(); b.run(); b.download()
    if args.skip_h2d_loop:
        elapsed_h2d, kms_h2d = None, [float("nan")]
    else:
        elapsed_h2d, kms_h2d = run_steps(args.warmup, args.steps, True, prime=True)   # host arrays in -> host results out (also: one pass per batch object, so every pinned / device buffer exists and every batch is resident)
    for k_ in dl_acc:
        dl_acc[k_] = 0
    elapsed, kms = run_steps(args.warmup, args.steps, False)           # THE timed region: W untimed + K timed steps over inputs resident in HBM
    dl = {k_: (dl_acc[k_] / max(1, dl_acc["n"])) for k_ in ("wait_kernel", "status_retries", "copy", "decode", "bytes_to_host")}

    batch = batches[0]
    res = batch.result()
    n_bad = int((res.status != 0).sum())
    in_b, out_b = batch.algorithmic_bytes()
    k_ms = float(np.mean(kms))
    info = batch.class_info(1)
    iters = int(batch.iterations().astype(np.int64).sum())
    pmc = committed_pmc(args)
    traffic = pmc.get("traffic_bytes_per_launch")

    if rank == 0:
        value_resident = args.graphs * world * args.steps / elapsed
        # the metric is H2D + kernel + D2H inclusive.  (A profiling run with --skip-h2d-loop has no such loop: it reports
        # the resident-input rate and says so in `value_is`.)
        el_value = elapsed_h2d if elapsed_h2d else elapsed
        value = args.graphs * world * args.steps / el_value
        achieved = (in_b + out_b) / (k_ms / 1e3) / 1e9
        step_text = ("decomposition kernels (exon join of every path into its record + result index written by the kernel) + D2H of status / records (paths and transcripts) / index + decode into the host path table with coverage = log(1 + weight)"
                     + (" + RCCL gather of the finished transcripts to rank 0 (device stream -> xGMI -> HBM of rank 0; rank 0's merge of the gathered streams into one transcript_set is NOT inside the step: it is host work at ~2.3 M bundles/s, profiles/r04/m_funnel_rehearsal_w8.txt)" if dist_on else "") + "; kernel k+1 is launched before the results of batch k are downloaded, so copies and decode run under the next kernel")
        line = {
            "metric": "bundles/sec", "value": value, "unit": "bundles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el_value / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "source": source_stamp(),
            "value_is": "h2d_inclusive" if elapsed_h2d else "resident (--skip-h2d-loop: profiling run)",
            "timed_step": ("H2D + kernel + D2H inclusive (SURVEY.md 8d): ald_batch_add_packed into the pinned host arrays of the batch + H2D of every array, three-stage pipeline on three host threads over 4 rotating batch objects -- K kernels, K downloads and K stagings between the brackets; the batches of the first timed steps are resident in HBM when the window opens (staged during the warm-up), later ones are staged under the kernels; " if elapsed_h2d else "inputs resident in HBM; ") + step_text,
            "value_resident": value_resident, "ms_per_step_resident": elapsed / args.steps * 1e3,
            "resident_step": "the same step over inputs that stay resident in HBM (nothing staged inside the window): " + step_text,
            "download_ms": {"wait_for_kernel": dl["wait_kernel"], "status_and_retries": dl["status_retries"], "d2h_copies": dl["copy"], "decode_paths_and_transcripts": dl["decode"],
                            "bytes_to_host_per_step": dl["bytes_to_host"]},
            "config": {"workload": f"{args.graphs} synthetic splice graphs per GPU, {args.vertices} vertices / {args.edges} edges each "
                                   + ("(BASELINE.json configs[1]; U[1,100) FP64 weights, 1 supporting sample per edge, no phasing paths)" if args.weights == "uniform" else
                                      f"(BASELINE.json configs[1] shape with {args.weights} weights -- secondary distribution, not the headline)"),
                       "graphs_per_gpu": args.graphs, "sharding": "independent graphs per rank, RCCL gather of finished transcripts to rank 0" if world > 1 else "single GPU",
                       "failed_graphs": n_bad, "paths_per_graph": float(len(res.weight)) / max(1, args.graphs),
                       "workgroups_per_cu": info["blocks_per_cu"], "grid": info["blocks_last_run"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": pmc.get("source"),
                         "traffic_gbs": (traffic / (k_ms / 1e3) / 1e9) if traffic else None,      # measured HBM traffic over this run's kernel time
                         "kernel": "ald_decomp_kernel_c1", "kernel_ms": k_ms, "kernel_ms_min": float(np.min(kms)), "kernel_ms_median": float(np.median(kms)), "kernel_ms_h2d_loop": float(np.mean(kms_h2d)),
                         "algorithmic_bytes_per_launch": in_b + out_b, "bytes_per_graph": (in_b + out_b) / args.graphs,
                         "kernel_graphs_per_s": args.graphs / (k_ms / 1e3),
                         # what actually bounds this kernel (dependent mutate-and-rescan steps, not bytes)
                         "steps_per_graph": iters / max(1, args.graphs), "steps_per_s_per_cu": iters / (k_ms / 1e3) / N_CUS,
                         "instr_per_graph": pmc.get("instr_per_graph"), "lds_bytes_per_workgroup": pmc.get("lds_bytes_per_workgroup")},
        }
        if world == 1 and not args.no_secondary:
            # the other single-GPU shapes of BASELINE.json / SURVEY.md 8d, kernel time only (parity for them: tests/test_gpu_parity.py)
            sec = {}
            cfg3 = A.synth(seed=1003, n_graphs=10000, v_min=8, v_max=512, edges_per_vertex=4)
            ms3, bad3, cls3, alg3 = kernel_only(A, dev, cfg3)
            sec["cfg3_mixed"] = {"workload": "BASELINE.json configs[2]: 10000 graphs, V ~ U{8..512}, E = 4V, seed 1003", "kernel_ms": ms3,
                                 "bundles_per_s": 10000 / (ms3 / 1e3), "failed_graphs": bad3, "graphs_per_class": cls3,
                                 "roofline": secondary_roofline(alg3, ms3, 10000, "ald_decomp_kernel_c0..c12 (one kernel per size class, three streams)")}
            flow = None
            if args.weights == "uniform":
                flow = A.synth(seed=seed, n_graphs=args.graphs, v_min=args.vertices, v_max=args.vertices, fixed_edges=args.edges, weight_mode=2)
                msf, badf, clsf, algf = kernel_only(A, dev, flow)
                sec["flow_weights"] = {"workload": f"{args.graphs} x {args.vertices}v/{args.edges}e, flow-conserving weights (SURVEY.md 8d's second distribution)",
                                       "kernel_ms": msf, "bundles_per_s": args.graphs / (msf / 1e3), "failed_graphs": badf, "graphs_per_class": clsf,
                                       "roofline": secondary_roofline(algf, msf, args.graphs, "ald_decomp_kernel_c1")}
            if args.cpu_sample > 0:                          # the port on the host cores for the two shapes, bounded samples (about 8 s each)
                cores_all = cpu_cores()
                sec["cfg3_mixed"]["cpu_baseline"] = secondary_cpu_baseline(cfg3, cores_all)
                if flow is not None:
                    sec["flow_weights"]["cpu_baseline"] = secondary_cpu_baseline(flow, cores_all)
            if elapsed_h2d:
                # row f1: the same host-arrays-in loop with every graph handed over RAW (ald_batch_add_packed_raw: extend_strands, boundary
                # grouping, phase projection, hyper_set ctor and filter_nodes run in the wave that loads the graph)
                el_raw, kms_raw = run_steps(2, args.steps, True, raw=True)
                sec["raw_form"] = {"workload": "the headline workload handed over as raw graphs (pre-steps of assembler::assemble on the device), host arrays in -> host results out",
                                   "bundles_per_s": args.graphs * args.steps / el_raw, "ms_per_step": el_raw / args.steps * 1e3, "kernel_ms": float(np.mean(kms_raw)),
                                   "vs_packed_form": (args.graphs * args.steps / el_raw) / (args.graphs * args.steps / elapsed_h2d)}
            sec["end_to_end_with_sink"] = sink_pipeline(A, pg, args.graphs)
            sec["dispatcher"] = dispatcher_rate()
            line["secondary"] = sec
        if args.cpu_sample > 0 and world == 1:           # reported at N = 1 only
            line["cpu_baseline"], line["cpu_baseline_1t"] = cpu_baseline_legs(pg, args.cpu_sample)
        emit(line)
    for b in batches:
        b.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

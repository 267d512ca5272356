#!/usr/bin/env python3
"""Decode-throughput benchmark of the MI355X DINT path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N > 1: starts one process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the decode kernel over the rank's whole resident
collection (every unit of every posting list; one GPU, default workload: 5e9
distinct postings, generated, encoded and uploaded in pieces during set-up). Inputs (encoded stream, unit
table, dictionary) are in HBM before the timed region starts. Posting lists
are partitioned statically across ranks, the dictionary is replicated, and
there is no data-path collective: the only collectives are the dictionary
broadcast during set-up and the max-over-ranks of the elapsed time. The two big
buffers (the encoded stream, the output) are chosen during set-up among
--placement-trials (6) candidates each, ranked by the library's own call for that
(dint_unit_table_rank_outputs): the kernel's time depends by 10-17 % on where
the driver puts the pair; what the process's FIRST
allocation reaches is reported next to it (value_first_allocation,
roofline.frac_first_allocation). --placement-trials 1: the first allocation only.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # config 2 (Gov2-shaped: 25 M documents) and config 4 (ClueWeb09-shaped: 50 M documents,
    # 1e10 postings over 8 GPUs = 1.25e9 per GPU)
    # gov2, one GPU: 5e9 DISTINCT postings (generated and encoded in chunks of --chunk-postings, set-up ~ 1.5 min). With more
    # ranks the host's cores are shared between them during set-up: 1e9 distinct postings per GPU, decoded x5 from five
    # device-side copies at distinct addresses — the same integers and bytes per step and GPU.
    "gov2": dict(universe=25_000_000, postings=5.0e9, replicate=1, multi_rank=dict(postings=1.0e9, replicate=5)),
    "clueweb": dict(universe=50_000_000, postings=1.25e9, replicate=1),
    # Workload-sensitivity rows (round 6, DESIGN.md section 4e: where the headline's roofline fraction holds and where not) —
    # secondary lines, 2e9 (freqs: 1e9) postings per GPU:
    # the docs generator turned up to the space the reference publishes for Gov2 docIDs (README.md:112-113: 5.94 bits per integer;
    # the default setting lands at 4.5): shorter clusters, flatter gap distributions
    "gov2-bpi59": dict(universe=25_000_000, postings=2.0e9, replicate=1,
                       synth=dict(stay_cluster=0.78, p_cluster_min=0.38, p_cluster_max=0.88)),
    # an exception-heavy stream: lists of at most 500 k postings in a universe of 2e9 documents — most gaps are in no dictionary
    "gov2-exceptions": dict(universe=2_000_000_000, postings=2.0e9, replicate=1, synth=dict(max_len=500_000)),
    # a .freqs-shaped stream (vroom_env/encode.cpp:160-164, jobs.hpp:74-84: the values are freq - 1, no prefix sum, the lists of
    # the docs file): per list a geometric law whose parameter is drawn in [0.17, 0.67) — mostly zeros in long runs for the
    # lists drawn high, the README's 3.05 bits per integer over all (README.md:114)
    "gov2-freqs": dict(universe=25_000_000, postings=1.0e9, replicate=1, values="freqs", freq_p=(0.17, 0.5)),
}
KERNEL_BY_TYPE = {"single_rect_dint": "decode_single_kernel", "single_packed_dint": "decode_single_kernel",
                  "multi_packed_dint": "decode_multi_kernel"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)  # (the shader clock is still ramping for the first few launches)
    ap.add_argument("--type", default="single_packed_dint", choices=sorted(KERNEL_BY_TYPE))
    ap.add_argument("--workload", default="gov2", choices=sorted(WORKLOADS),
                    help="gov2: universe 25M, 5e9 distinct postings encoded per GPU (SURVEY §8d config 2; with --gpus > 1: 1e9 per GPU "
                         "decoded x5 from distinct addresses); clueweb: universe 50M, 1.25e9 postings per GPU (config 4 at 8 GPUs)")
    ap.add_argument("--chunk-postings", type=float, default=1.0e9,
                    help="set-up generates, encodes and uploads the rank's shard in pieces of about this many postings (host memory: "
                         "one piece at a time)")
    ap.add_argument("--postings", type=float, default=None, help="postings encoded per GPU (weak scaling: fixed per GPU)")
    ap.add_argument("--placement-trials", type=int, default=6,
                    help="N (default 6; until round 5: 4 — on one of that round's boxes the first three candidate outputs were all the slow kind) candidate output buffers, ranked by the library (dint_unit_table_rank_outputs), then N candidate "
                         "copies of the stream, allocated during set-up; the pair the decode kernel runs fastest on is kept (the kernel's "
                         "time differs by 10-17 %% with WHERE the driver puts the two buffers: DESIGN.md section 4e). 1: the process's "
                         "first allocation, no selection — the fast level in nine fresh processes in a row on some boxes "
                         "(profiles/r04_first_allocation.txt), the slow one on another (profiles/r04_bench_first_allocation_slow_box.json); "
                         "the line reports the first allocation's time either way (value_first_allocation, roofline.frac_first_allocation)")
    ap.add_argument("--place-by-probe", action="store_true",
                    help="placement by the library's SAMPLED probe (dint_probe_placement) instead of full-size candidates: three copies of "
                         "the stream (an eighth of the output each) allocated at three points of the set-up — with the uploaded pieces, "
                         "in front of the output, behind it — are each decoded against the ONE output buffer over an evenly spread "
                         "sample of the unit table (--probe-ints integers, four launches), the fastest copy stays, the others are freed: "
                         "tens of milliseconds and a quarter of the working set in trial memory, against --placement-trials 6's twelve "
                         "full-size buffers. Use with --placement-trials 1")
    ap.add_argument("--probe-ints", type=float, default=4.0e8, help="integers the placement probe's sample decodes per launch")
    ap.add_argument("--probe-eval", action="store_true",
                    help="with --placement-trials N > 1: run the sampled probe over the same candidates as well and report both "
                         "rankings (config.placement_probe_eval): does the sample see what the full-size launches see?")
    ap.add_argument("--apart-gb", type=float, default=24.0,
                    help="the stream and the output are not allocated next to each other: this much device memory is allocated between "
                         "them and freed again (two big buffers allocated one after the other usually land in the same kind of physical "
                         "stretch, the slow placement of DESIGN.md section 4e). 0: no spacer")
    ap.add_argument("--multi-rank-distinct", default="auto", choices=["auto", "on", "off"],
                    help="--gpus N > 1, gov2: `on` = every rank generates, encodes and decodes the SAME workload as a one-GPU run "
                         "(5e9 distinct postings per GPU: N=1 and N>1 lines differ in the rank count alone); `off` = 1e9 distinct "
                         "postings per GPU decoded x5 from five device-side copies (the same integers and bytes per step; set-up on "
                         "a host whose cores the ranks share is 5x shorter); `auto` (default) = on when the container has at least "
                         "8 CPUs per rank (cgroup quota / affinity), else off. The line says which: distinct_postings_per_gpu, replicate")
    ap.add_argument("--replicate", type=int, default=None,
                    help="device-side copies of the encoded shard at distinct addresses, all decoded in one step")
    ap.add_argument("--universe", type=int, default=None, help="documents")
    ap.add_argument("--unit-ints", type=int, default=16384,
                    help="the sidecar's granularity: integers per unit (16384: 1.6 %% faster than 8192 at 5e9 integers per launch, profiles/r03_unit_sweep.txt)")
    ap.add_argument("--dict-sample", type=float, default=2.0e7,
                    help="postings the DSF dictionary statistics are collected from")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="decode time budget of each leg of the CPU baseline sample (0 = skip)")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="JSON written by tools/pmc_traffic.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS command "
                         "(HBM counters cannot be read from inside the run). Default: the round's committed measurement, attached only "
                         "if it was taken for the same type, the same integers per launch AND the same build of libdint_hip.so "
                         "(its lib_sha16 = this process's library); otherwise roofline.traffic is null and the note says why")
    ap.add_argument("--no-verify", action="store_true", help="skip the full bit-exact output check")
    ap.add_argument("--per-launch-schedule", action="store_true",
                    help="decode through dint_decode_units (the bundle schedule rebuilt before every launch, and timed) instead "
                         "of a unit table prepared once during set-up")
    ap.add_argument("--as-rank", default=None, metavar="K/W",
                    help="one process, one GPU, decoding exactly the shard rank K of a W-rank job would get "
                         "(sharding.partition_lists(lens_all, W)[K] of the W x postings collection): BASELINE config 4's "
                         "1-of-8 shards on a single-GPU box. The line says `emulated_rank`; it is not a scaling measurement")
    ap.add_argument("--rank-timeout", type=float, default=3600.0,
                    help="--gpus N launched from this process: seconds a rank may run before all of them are stopped")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise torch.distributed (RCCL on a GPU) even with one rank, and run the reductions through it")
    return ap.parse_args(argv)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` on its own: start N fresh processes, one per GPU, and relay rank 0's line.
    (This process has not touched the GPU; the children are new processes, not a re-exec.) When a rank fails or
    the time runs out, the others are stopped (they would wait in a collective for ever) and the exit code is
    not zero."""
    import signal
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if rank == 0 else subprocess.DEVNULL, start_new_session=True))
    def stop_all():  # exactly the process groups started above
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                p.wait()

    # The ranks run in sessions of their own (a rank that hangs in a collective is stopped by process group), so a signal to
    # THIS process no longer reaches them: SIGTERM / SIGINT here (a `timeout` wrapper, Ctrl-C) stop them first.
    def on_signal(signum, _frame):
        stop_all()
        signal.signal(signum, signal.SIG_DFL)
        os.kill(os.getpid(), signum)

    old_handlers = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    deadline = time.monotonic() + args.rank_timeout
    failed = None
    try:
        while failed is None and any(p.poll() is None for p in procs):
            for rank, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = f"rank {rank} exited with code {p.returncode}"
                    break
            else:
                if time.monotonic() > deadline:
                    failed = f"no result after {args.rank_timeout:.0f}s"
                else:
                    time.sleep(0.2)
        if failed is None:
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
    finally:
        if failed is not None or any(p.poll() is None for p in procs):  # (the second: an exception on the way)
            stop_all()
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    if failed is not None:
        print(f"[bench] {failed}: all ranks stopped", file=sys.stderr, flush=True)
        return 1
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return 0


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores() -> int:
    """Distinct (socket, core) pairs among the CPUs this process may run on."""
    try:
        allowed = os.sched_getaffinity(0)
        seen = set()
        for cpu in allowed:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            with open(base + "physical_package_id") as f, open(base + "core_id") as g:
                seen.add((f.read().strip(), g.read().strip()))
        return len(seen) or len(allowed)
    except OSError:
        return len(os.sched_getaffinity(0))


def lib_sha16():
    """first 16 hex digits of the sha256 of the HIP library this process runs (what a traffic file must have been measured on)"""
    import hashlib

    path = os.environ.get("DINT_HIP_LIB") or os.path.join(ROOT, "dint_amd", "libdint_hip.so")
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max, v1 cfs quota), or None: no limit."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        return None if q == "max" else float(q) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, period = float(f.read()), float(g.read())
        return None if q <= 0 else q / period
    except (OSError, ValueError):
        return None


def cpu_baseline(kind, dict_file, enc, list_byte_starts, seconds):
    """The oracle's restatement of the reference decode, timed on this host: (i) one thread, per-list timing
    summed exactly like vroom_env/decode.cpp:139-150; (ii) every core, the lists statically partitioned by
    stream bytes, wall time. A bounded sample: each leg stops after about `seconds` of decode time."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # the CPU restatement, used here as the timed baseline only

    od = oracle.OracleDict(kind, dict_file)
    sec = ints = lists = passes = 0
    while sec < seconds:
        s1, i1, l1 = od.time_stream(enc, max_seconds=seconds - sec)
        sec, ints, lists, passes = sec + s1, ints + i1, lists + l1, passes + 1
    one = {"value": round(ints / sec / 1e6, 2), "unit": "M ints/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
           "sample": f"{passes} pass(es) over the same encoded stream, {lists} list decodes ({ints} postings), per-list "
                     f"timing summed as in vroom_env/decode.cpp:139-150, {sec:.1f}s of decode time"}
    # "every core" = every CPU this process may use: its affinity mask, cut down to the container's CPU-time quota
    # (cgroup cpu.max) — more runnable threads than that are throttled, which measures the throttle, not the host
    affinity, quota = len(os.sched_getaffinity(0)), cpu_quota()
    threads = affinity if quota is None else max(1, min(affinity, int(quota)))
    phys = min(physical_cores(), threads)
    # thread k takes the lists that begin in [k, k + 1) / threads of the stream's bytes (a thread may get none): one
    # pthread each inside liboracle, a persistent decode buffer per thread, all looping until `seconds` have passed
    cut = np.searchsorted(list_byte_starts, [enc.size * k // threads for k in range(1, threads)])
    starts = [0] + [int(list_byte_starts[i]) if i < len(list_byte_starts) else enc.size for i in cut]
    wall, tot, nlists = od.time_stream_parallel(enc, starts, seconds)
    one["all_cores"] = {"value": round(tot / wall / 1e6, 2), "unit": "M ints/s", "cores": phys, "threads": threads,
                        "host_logical_cpus": affinity, "host_physical_cores": physical_cores(), "cgroup_cpu_quota": quota,
                        "speedup_over_one_core": round(tot / wall / (ints / sec), 1),
                        "sample": f"{threads} pthreads (the container's CPU quota: {quota if quota is not None else 'none'}; the host has "
                                  f"{physical_cores()} physical cores / {affinity} threads), contiguous list ranges of equal stream bytes, "
                                  f"each looping over its range with its own reused buffer for {seconds:.0f}s: {nlists} list "
                                  f"decodes, {tot} postings in {wall:.2f}s wall (first thread start to last thread end)"}
    return one


def main():
    args = parse_args()
    w = dict(WORKLOADS[args.workload])
    n_ranks = max(args.gpus, int(os.environ.get("WORLD_SIZE", "1")))
    if args.as_rank is not None:
        w.update(w.get("multi_rank", {}))  # (one emulated shard after the other on one GPU: the short set-up)
    elif n_ranks > 1:
        # the ranks share the host's cores during set-up (generation + encode): the one-GPU workload as it is when every
        # rank has at least 8 CPUs to itself, else a fifth of it decoded from five copies (same integers and bytes per step)
        affinity, quota = len(os.sched_getaffinity(0)), cpu_quota()
        cpus = affinity if quota is None else min(affinity, quota)
        distinct = args.multi_rank_distinct == "on" or (args.multi_rank_distinct == "auto" and cpus / n_ranks >= 8)
        if not distinct:
            w.update(w.get("multi_rank", {}))
    postings = int(args.postings if args.postings is not None else w["postings"])
    R = max(1, args.replicate if args.replicate is not None else w["replicate"])
    universe = args.universe if args.universe is not None else w["universe"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))  # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # the shard this process decodes: its own rank's — or, --as-rank K/W, the one rank K of a W-rank job would get
    shard_rank, shard_world = rank, world
    if args.as_rank is not None:
        if world != 1:
            raise SystemExit("--as-rank emulates one rank of a larger job in ONE process (--gpus 1)")
        try:
            shard_rank, shard_world = (int(x) for x in args.as_rank.split("/"))
        except ValueError:
            raise SystemExit("--as-rank wants K/W, e.g. 3/8")
        if not 0 <= shard_rank < shard_world:
            raise SystemExit("--as-rank K/W: 0 <= K < W")

    import numpy as np
    import torch

    stub = os.environ.get("DINT_BENCH_STUB")  # tests only: a module that stands in for the device layer (tests/bench_stub.py)
    if stub:
        import importlib

        device = importlib.import_module(stub)
        backend, dev = "gloo", torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
        from dint_amd import device

        torch.cuda.set_device(local_rank)
        backend, dev = "nccl", torch.device("cuda", local_rank)
    from dint_amd import host, sharding

    distributed = world > 1 or args.force_process_group
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (one rank on its own, --force-process-group)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))

    kind = host.KIND_BY_TYPE[args.type]
    threads = max(1, host.default_threads() // world)

    # ---- set-up (untimed): collection shard, dictionary, encode, upload ----------
    t0 = time.time()
    p = host.synth_params(universe=universe, seed=args.seed, **w.get("synth", {}))
    # The collection is `world * postings` postings; list lengths are drawn once
    # (same on every rank) and contiguous list ranges balanced by postings are
    # handed to the ranks.
    lens_all = host.synth_lengths(p, postings * shard_world)
    lo, hi = sharding.partition_lists(lens_all, shard_world)[shard_rank]
    lens = lens_all[lo:hi]
    n_shard = int(lens.sum(dtype=np.uint64))
    if n_shard == 0:
        raise SystemExit(f"shard {shard_rank} of {shard_world} is empty: {postings * shard_world} postings are too few for "
                         f"{shard_world} ranks (the longest list alone holds {int(lens_all.max())})")
    # pieces of the shard: contiguous list ranges of about --chunk-postings postings. A list's gaps depend on the seed and
    # the list's ordinal alone (synth_gaps' first_list_id), so the pieces ARE the shard, generated one at a time.
    cum = np.cumsum(lens, dtype=np.uint64)
    n_pieces = max(1, int(round(n_shard / max(1.0, args.chunk_postings))))
    cuts = [0] + [int(np.searchsorted(cum, n_shard * k // n_pieces, side="left")) + 1 for k in range(1, n_pieces)] + [len(lens)]
    cuts = sorted(set(min(c, len(lens)) for c in cuts))

    def piece(i):
        a, b = cuts[i], cuts[i + 1]
        if w.get("values") == "freqs":
            # freq - 1 of every posting of the lists [a, b): a function of the seed and the piece's first list alone
            r = np.random.default_rng([args.seed, lo + a])
            p_lo, p_span = w["freq_p"]
            p_list = np.repeat(p_lo + p_span * r.random(b - a), lens[a:b])
            return host.Collection((r.geometric(p_list) - 1).astype(np.uint32), lens[a:b])
        return host.Collection(host.synth_gaps(p, lens[a:b], first_list_id=lo + a, threads=threads), lens[a:b])

    coll0 = piece(0)
    log(rank, f"rank shard: lists [{lo},{hi}) = {n_shard} postings in {len(cuts) - 1} piece(s); the first generated in {time.time() - t0:.1f}s")

    def build_dictionary(sample_of):
        # n-gram counting and selection on the device (byte-identical to the host construction, 7 s -> 0.2 s of set-up)
        if hasattr(device, "build_dictionary"):
            return device.build_dictionary(kind, sample_of, max_sample_ints=int(args.dict_sample), device=local_rank)[0]
        return host.build_dictionary(kind, sample_of, max_sample_ints=int(args.dict_sample), threads=threads)

    t0 = time.time()
    if rank == 0 and shard_rank != 0:
        # --as-rank K/W, K != 0: the job's dictionary comes from RANK 0's first lists, not from this shard's —
        # regenerate that prefix sample (the lists build_dictionary would take from rank 0's shard)
        hi0 = sharding.partition_lists(lens_all, shard_world)[0][1]
        cum0 = np.cumsum(lens_all[:hi0], dtype=np.uint64)
        j = max(1, int(np.searchsorted(cum0, int(args.dict_sample), side="right"))) if args.dict_sample else hi0
        sample = host.Collection(host.synth_gaps(p, lens_all[:j], first_list_id=0, threads=threads), lens_all[:j])
        dict_file = build_dictionary(sample)
        del sample
    elif rank == 0:
        # dictionary statistics from a prefix sample of the collection (rank 0's first lists)
        dict_file = build_dictionary(coll0)
    else:
        dict_file = None
    if distributed:
        box = [dict_file]
        dist.broadcast_object_list(box, src=0)
        dict_file = box[0]
    log(rank, f"dictionary: {len(dict_file)} B in {time.time() - t0:.1f}s")

    # every piece: encode on the host, upload the stream, the unit table (offsets moved to the piece's place in the shard) and
    # — for the bit-exact check at the end — the gaps themselves (the expected output stays on the device)
    d = device.Dictionary(kind, dict_file, device=local_rank)
    info = d.info()
    t0 = time.time()
    enc_parts, unit_parts, expect_parts = [], [], []
    enc_bytes_one = ints_done = lists_done = 0
    enc = units = None  # the FIRST piece's stream and unit table stay on the host: the CPU baseline's sample
    for i in range(len(cuts) - 1):
        c = coll0 if i == 0 else piece(i)
        e_i, u_i = host.encode_vroom(kind, dict_file, c, unit_ints=args.unit_ints, threads=threads)
        u_i = u_i.copy()
        if i == 0:
            enc, units = e_i, u_i.copy()
        u_i["in_off"] += np.uint64(enc_bytes_one)
        u_i["out_off"] += np.uint64(ints_done)
        u_i["list"] += np.uint32(lists_done)
        enc_parts.append(torch.from_numpy(e_i).to(dev))
        unit_parts.append(u_i)
        if not args.no_verify:
            expect_parts.append(torch.from_numpy(np.ascontiguousarray(c.gaps).view(np.int32)).to(dev))
        enc_bytes_one += e_i.size
        ints_done += c.num_postings
        lists_done += len(c.lens)
        log(rank, f"piece {i}: {c.num_postings} postings -> {e_i.size} B, {len(u_i)} units ({time.time() - t0:.1f}s)")
        del c, e_i
    del coll0
    assert ints_done == n_shard
    bpi = enc_bytes_one * 8 / n_shard
    units_one = np.concatenate(unit_parts)
    del unit_parts
    log(rank, f"encoded: {enc_bytes_one} B ({bpi:.3f} bits/int), {len(units_one)} units in {time.time() - t0:.1f}s")
    expect_dev = torch.cat(expect_parts) if len(expect_parts) > 1 else (expect_parts[0] if expect_parts else None)
    del expect_parts

    n_ints = n_shard * R
    enc_one = torch.cat(enc_parts) if len(enc_parts) > 1 else enc_parts[0]
    del enc_parts
    units_all = np.tile(units_one, R)
    for r in range(R):
        sl = slice(r * len(units_one), (r + 1) * len(units_one))
        units_all["in_off"][sl] += np.uint64(r * enc_bytes_one)
        units_all["out_off"][sl] += np.uint64(r * n_shard)
    units_dev = device.units_to_device(units_all, dev)
    n_units = len(units_all)

    def allocate_stream():
        e = torch.empty(enc_bytes_one * R, dtype=torch.uint8, device=dev)
        for r in range(R):
            e[r * enc_bytes_one:(r + 1) * enc_bytes_one].copy_(enc_one)
        return e

    def allocate_output():
        return torch.empty(n_ints, dtype=torch.int32, device=dev)

    # Placement (set-up, untimed): the decode kernel's time depends on WHERE the driver puts the stream it reads and the
    # output it writes — 10-17 % between two pairs of buffers in one process, stable for the life of a pair, and nothing
    # a plain fill / copy / gather notices (tools/archive/box_spread/realloc_probe.py). A caller who keeps its
    # buffers for many decodes picks them once; so does the bench: a few candidate output buffers for the first copy of
    # the stream, then a few candidate copies of the stream for the output buffer that won, two launches each; the
    # fastest pair stays, the others are freed before the timed region.
    trials = max(1, args.placement_trials) if dev.type == "cuda" and not os.environ.get("DINT_BENCH_STUB") else 1
    by_probe = bool(args.place_by_probe) and dev.type == "cuda" and not os.environ.get("DINT_BENCH_STUB") and hasattr(device, "probe_placement")
    enc_dev = allocate_stream() if (R > 1 or trials > 1 or by_probe) else enc_one  # (one copy, no candidates: the uploaded stream itself)
    # the output is allocated APART from the stream: a spacer between the two allocations, freed right away
    spacer = None
    if dev.type == "cuda" and args.apart_gb > 0:
        try:
            free_b = torch.cuda.mem_get_info(dev)[0]
            want = int(min(args.apart_gb * 1e9, max(0, free_b - 4 * n_ints - (8 << 30))))
            if want > (1 << 30):
                spacer = torch.empty(want, dtype=torch.uint8, device=dev)
        except RuntimeError:
            spacer = None
    out_dev = allocate_output()
    del spacer
    placement_ms = None
    probe_report = None
    if by_probe:
        # three copies of the stream, allocated at three points of the set-up: with the uploaded pieces (enc_one: R == 1 only),
        # in front of the output (enc_dev), behind the output
        cands = ([enc_one] if R == 1 else []) + [enc_dev]
        try:
            cands.append(allocate_stream())
        except RuntimeError:
            pass
        torch.cuda.synchronize(dev)
        t_probe = time.perf_counter()
        pm = device.probe_placement(d, cands, units_dev, n_units, [out_dev], int(args.probe_ints))[:, 0]
        probe_s = time.perf_counter() - t_probe
        # (what the full-size launch reaches on the set-up's first pair — the stream copy in front of the output — and on every
        # candidate: two launches each, for the report only)
        full = []
        for e in cands:
            ms2 = []
            for _ in range(2):
                d.decode_units(e, units_dev, n_units, out_dev)
                torch.cuda.synchronize(dev)
                ms2.append(d.last_kernel_ms())
            full.append(round(min(ms2), 4))
        keep = int(np.argmin(pm))
        probe_report = {"stream_copies": len(cands), "sample_ints": int(args.probe_ints), "sampled_kernel_ms": [round(float(x), 4) for x in pm],
                        "full_size_kernel_ms_of_the_same_copies": full, "kept": keep, "probe_seconds": round(probe_s, 3),
                        "first_pair": (1 if R == 1 else 0),
                        "trial_memory_over_working_set": round((len(cands) - 1) * enc_bytes_one * R / (enc_bytes_one * R + 4 * n_ints), 3)}
        log(rank, f"placement probe: {probe_report}")
        enc_dev = cands[keep]
        del cands
        torch.cuda.empty_cache()

    def rank_candidates(e, outs):
        # the library's own call for this (include/dint_hip.h: dint_unit_table_rank_outputs): every candidate decoded three
        # times through a prepared unit table, the faster of the last two launches' kernel times each
        if not args.per_launch_schedule and hasattr(device, "UnitTable") and hasattr(device.UnitTable, "rank_outputs"):
            tab = device.UnitTable(d, e, units_dev, n_units, n_ints)
            ms, _ = tab.rank_outputs(outs)
            tab.close()
            return [round(x, 4) for x in ms]
        res = []
        for o in outs:
            ms = []
            for _ in range(3):
                d.decode_units(e, units_dev, n_units, o)
                torch.cuda.synchronize(dev)
                ms.append(d.last_kernel_ms())
            res.append(round(min(ms[1:]), 4))
        return res

    def candidates(first, allocate):
        out = [first]
        for _ in range(trials - 1):
            try:
                out.append(allocate())
            except RuntimeError:  # (out of memory: fewer candidates)
                break
        return out

    probe_eval = None
    if trials > 1:
        outs = candidates(out_dev, allocate_output)
        ms_out = rank_candidates(enc_dev, outs)
        if args.probe_eval and hasattr(device, "probe_placement"):
            probe_eval = {"sample_ints": int(args.probe_ints),
                          "output_buffers_sampled_ms": [round(float(x), 4) for x in device.probe_placement(d, [enc_dev], units_dev, n_units, outs, int(args.probe_ints))[0]]}
        out_dev = outs[int(np.argmin(ms_out))]
        del outs
        torch.cuda.empty_cache()
        encs = candidates(enc_dev, allocate_stream)
        ms_enc = [min(ms_out)] + [rank_candidates(e, [out_dev])[0] for e in encs[1:]]
        if probe_eval is not None:
            probe_eval["stream_buffers_sampled_ms"] = [round(float(x), 4) for x in device.probe_placement(d, encs, units_dev, n_units, [out_dev], int(args.probe_ints))[:, 0]]
            probe_eval["output_buffers_full_ms"], probe_eval["stream_buffers_full_ms"] = ms_out, ms_enc
        enc_dev = encs[int(np.argmin(ms_enc))]
        del encs
        torch.cuda.empty_cache()
        placement_ms = {"output_buffers": ms_out, "stream_buffers": ms_enc}
        log(rank, f"placement: kernel ms of the candidate output buffers {ms_out}, of the candidate stream buffers {ms_enc}")
    del enc_one
    end_dev = torch.zeros(n_units, dtype=torch.int64, device=dev)
    log(rank, f"device buffers: enc {enc_dev.data_ptr():#x} ({enc_dev.data_ptr() % (2 << 20):#x} mod 2 MB) "
              f"out {out_dev.data_ptr():#x} units {units_dev.data_ptr():#x}")

    def sync_all():
        if distributed:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    # The unit table is prepared once (set-up, untimed — like the sidecar it is a property of the encoded collection,
    # SURVEY H3): which tiny units share a tile, the work items of the unit queue. A step is then ONE launch.
    # --per-launch-schedule: dint_decode_units instead, which rebuilds that schedule before every launch.
    unit_table = None
    if not args.per_launch_schedule and hasattr(device, "UnitTable"):
        unit_table = device.UnitTable(d, enc_dev, units_dev, n_units, n_ints)

    def step():
        if unit_table is not None:
            unit_table.decode(out_dev, end_dev)
        else:
            d.decode_units(enc_dev, units_dev, n_units, out_dev, end_dev)

    # ---- warm-up ------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    sync_all()

    # ---- timed region: exactly K steps (each returns the end offsets too, like the reference's decode) ----
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t_start
    # per-launch kernel time of the timed launches themselves: the event pairs the library recorded around them
    kernel_ms = np.asarray(d.recent_kernel_ms(min(args.steps, 64)), dtype=np.float64)
    # the clock the last timed launch actually ran at (its first wave's cycle count over its duration)
    shader_mhz = round(float(d.last_kernel_clock_mhz()), 1) if hasattr(d, "last_kernel_clock_mhz") else None
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_ints], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_ints = int(tot.item())
    else:
        total_ints = n_ints

    # ---- correctness: bit-exact against the encoder's input -------------------------
    ends = end_dev.cpu().numpy().view(np.uint64)
    payload_bytes = int((ends - units_all["in_off"]).sum())
    first_alloc_ms = placement_ms["output_buffers"][0] if placement_ms else None
    if probe_report is not None:
        first_alloc_ms = probe_report["full_size_kernel_ms_of_the_same_copies"][probe_report["first_pair"]]
    bit_exact = None
    if not args.no_verify:
        bit_exact = True
        for r in range(R):  # replica by replica, piece-sized slices, on the device (the expected gaps were uploaded during set-up)
            for a in range(0, n_shard, 1 << 28):
                b = min(n_shard, a + (1 << 28))
                bit_exact = bit_exact and bool(torch.equal(out_dev[r * n_shard + a:r * n_shard + b], expect_dev[a:b]))
        if distributed:
            ok = torch.tensor([1 if bit_exact else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            bit_exact = bool(ok.item())
        if not bit_exact:
            raise SystemExit("FATAL: decoded integers differ from the encoder's input")

    # ---- the box: what plain fill and copy kernels reach on the same output buffer, after the timed region (boxes of
    # one pool differ by 15 % in the decode kernel's time at equal clocks: this says whether their HBM does too) -----
    box_probe = None
    if dev.type == "cuda" and rank == 0:
        half = (n_ints // 2) & ~1023
        probe = {}
        for name, fn, nbytes in (("fill_gbps", lambda: out_dev.zero_(), 4 * n_ints),
                                 ("copy_gbps", lambda: out_dev[:half].copy_(out_dev[half:2 * half]), 8 * half)):
            if nbytes == 0:
                continue
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            probe[name] = round(3 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        # ... and random 4-byte gathers from a table that fits one XCD's L2 (2 MB) and from one that only the
        # memory-side cache holds (64 MB): the decode kernel's cold codewords are such gathers
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        for name, words in (("gather_2mb_gps", 1 << 19), ("gather_64mb_gps", 1 << 24)):
            table = torch.arange(words, dtype=torch.int32, device=dev)
            idx = torch.randint(0, words, (1 << 26,), dtype=torch.int64, device=dev, generator=g)
            torch.take(table, idx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                torch.take(table, idx)
            e1.record()
            torch.cuda.synchronize(dev)
            probe[name] = round(3 * idx.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del table, idx
        props = torch.cuda.get_device_properties(dev)
        probe["compute_units"] = int(props.multi_processor_count)
        probe["device"] = f"{props.name} {getattr(props, 'gcnArchName', '')}".strip()
        box_probe = probe

    if rank == 0:
        # ---- what the stream is made of (host pre-pass over one replica) -------------
        st = d.stream_stats(enc)  # (the first piece of the shard: a sample)
        slots = st.codewords + st.exceptions16 + st.exceptions32
        stream = {
            "ints_per_codeword": round(st.ints / max(1, slots), 3),
            "exception_pct": round(100.0 * (st.exceptions16 + st.exceptions32) / max(1, slots), 4),
            "exception32_pct": round(100.0 * st.exceptions32 / max(1, slots), 4),
            "lds_hit_pct": round(100.0 * st.hot_codewords / max(1, st.codewords), 2),
            "lds_hit_pct_of_ints": round(100.0 * st.hot_ints / max(1, st.ints), 2),
        }
        if kind == host.MULTI_PACKED:
            stream["blocks_16bit_pct"] = round(100.0 * st.wide_blocks / max(1, st.wide_blocks + st.narrow_blocks), 2)

        # ---- CPU baseline (N=1 only): the oracle timed like vroom_env/decode.cpp ------
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            first = np.r_[True, units["list"][1:] != units["list"][:-1]]
            # a list's header starts where the previous list's payload ended
            prev_end = np.r_[np.uint64(0), ends[: len(units)][np.flatnonzero(np.r_[first[1:], True])][:-1]]  # (the first piece's lists)
            cpu = cpu_baseline(kind, dict_file, enc, prev_end.astype(np.int64), args.cpu_seconds)

        algo_bytes = 4 * n_ints + payload_bytes  # per launch, this rank
        traffic, traffic_note = None, "not measured in this process (rocprofv3 --pmc passes: tools/profile_round.sh)"
        if args.traffic_file and os.path.exists(args.traffic_file):
            with open(args.traffic_file) as f:
                tf = json.load(f)
            lib_now = lib_sha16()
            if tf.get("type") == args.type and tf.get("ints_per_launch") == n_ints and tf.get("lib_sha16") != lib_now:
                traffic_note = (f"{os.path.relpath(args.traffic_file, ROOT)} was measured on another build of libdint_hip.so "
                                f"({tf.get('lib_sha16')}; this process runs {lib_now}): not attached; " + traffic_note)
            elif tf.get("type") == args.type and tf.get("ints_per_launch") == n_ints:
                traffic = {"write_gb": tf["write_gb"], "fetch_gb_raw": tf["fetch_gb_raw"],
                           "fetch_gb_corrected": tf["fetch_gb_corrected"],
                           "total_gb": round(tf["write_gb"] + tf["fetch_gb_corrected"], 3),
                           "total_bytes": int(round((tf["write_gb"] + tf["fetch_gb_corrected"]) * 1e9)),
                           "over_algorithmic": round((tf["write_gb"] + tf["fetch_gb_corrected"]) * 1e9 / algo_bytes, 4)}
                traffic_note = tf.get("note", "") + f" [{os.path.relpath(args.traffic_file, ROOT)}: separate profiler passes of this command, not this process]"
        k_mean = float(kernel_ms.mean())
        achieved = algo_bytes / (k_mean * 1e-3) / 1e9
        value = total_ints * args.steps / elapsed / 1e6
        line = {
            "metric": f"M ints/sec decoded (vroom {args.type})",
            "value": round(value, 1),
            "unit": "M ints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # what the same launches reach on the process's first allocation of the two big buffers (kernel time of the set-up
            # trial on it; `value` is the timed region on the pair the set-up kept: config.placement)
            "value_first_allocation": (round(n_ints / (first_alloc_ms * 1e-3) / 1e6 * world, 1) if first_alloc_ms else round(value, 1)),
            "dtype": "u32",
            "data": "synthetic" if not stub else "stub",
            **({"stub": stub, "stub_note": "the device layer was replaced by a test stand-in: nothing was decoded, value and "
                                           "roofline are not measurements"} if stub else {}),
            **({"emulated_rank": f"{shard_rank}/{shard_world}",
                "emulated_note": f"one process on one GPU decoding the shard rank {shard_rank} of a {shard_world}-rank job "
                                 f"would get (lists [{lo},{hi}) of {len(lens_all)}); not a scaling measurement"}
               if args.as_rank is not None else {}),
            "bit_exact": bit_exact,
            # what one step decodes on every GPU: `distinct_postings_per_gpu` different postings, `replicate` times over (from
            # that many device-side copies of the stream at distinct addresses); a one-GPU run is 5e9 x 1, a multi-rank run
            # on a host with few CPUs per rank 1e9 x 5 (--multi-rank-distinct) — the same integers and bytes per step either way
            "distinct_postings_per_gpu": n_shard,
            "replicate": R,
            "config": {
                # (nominal figures: the same text for every rank count but for the count itself; rank 0's exact shard is
                # distinct_postings_per_gpu)
                "workload": f"{args.type} decode, DSF-65536-16 dictionary (hot set in LDS), {args.workload}-shaped synthetic "
                            + ("term frequencies minus one (a .freqs stream: no prefix sum)" if w.get("values") == "freqs" else "docIDs")
                            + f": universe {universe}, {postings} distinct postings encoded per GPU"
                            + (f", generator settings {w['synth']}" if w.get("synth") else "")
                            + (f", decoded x{R} per step from {R} device-side copies at distinct addresses" if R > 1 else "")
                            + f", {shard_world} GPU(s): contiguous list ranges of one collection of {postings * shard_world} postings",
                "distinct_postings_per_gpu": n_shard,
                "ints_per_gpu_per_step": n_ints,
                "lists_per_gpu": int(np.count_nonzero(lens)) * R,
                "stream_statistics_sample": f"the first {int(st.ints)} postings of the shard",
                "units_per_gpu": n_units,
                "unit_ints": args.unit_ints,
                "schedule": "prepared unit table (set-up)" if unit_table is not None else "per launch (timed)",
                "placement": (f"fastest of {len(placement_ms['output_buffers'])} candidate output buffers, then of "
                              f"{len(placement_ms['stream_buffers'])} candidate stream buffers, chosen during set-up"
                              if placement_ms else
                              (f"sampled probe (dint_probe_placement): the fastest of {probe_report['stream_copies']} copies of the stream "
                               f"against the one output buffer, chosen during set-up in {probe_report['probe_seconds']} s") if probe_report
                              else "first allocation"
                              + (f" (output allocated {args.apart_gb:g} GB apart from the stream)" if args.apart_gb > 0 else "")),
                "placement_trial_kernel_ms": placement_ms,
                "placement_probe": probe_report,
                "placement_probe_eval": probe_eval,
                "bits_per_int": round(bpi, 3),
                **stream,
                "hot_codewords_in_lds": int(info.hot_entries),
                "lds_bytes": int(info.lds_bytes),
                "parallelism": f"list-range x{world}" if args.as_rank is None else f"list-range shard {shard_rank} of {shard_world}",
                "process_group": (backend if not stub else "gloo") if distributed else None,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # the same launch on the process's FIRST allocation of the two buffers, before any candidate was tried (set-up)
                "frac_first_allocation": (round(algo_bytes / (first_alloc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                          if first_alloc_ms else round(achieved / HBM_PEAK_GBS, 4)),
                "kernel_ms_first_allocation": first_alloc_ms if first_alloc_ms else round(k_mean, 4),
                "first_allocation_note": "the first CANDIDATE pair of the set-up: the first copy of the stream and the first output buffer, "
                                         "allocated behind the uploaded pieces, the expected gaps and a spacer (--apart-gb) — not the "
                                         "process's very first device allocation",
                "traffic": traffic,
                "traffic_note": traffic_note,
                # (a prepared multi-dictionary table of block-granular units: the kernel compiled without the unit queue, and behind
                # it, inside the same event pair, the same kernel again over the units that fit no tile — about one in a hundred,
                # cut in two records each when the table was prepared — and the general kernel for the few that could not be cut)
                "kernel": ("decode_multi_bundles_kernel (+ a second launch of it over the units that fit no tile, cut in two)"
                           # (units of several blocks: the prepared table finds the blocks itself, once — DINT_OPT_REFINE_UNITS —
                           # unless a unit holds more than 131072 integers)
                           if args.type == "multi_packed_dint" and unit_table is not None and args.unit_ints <= 131072
                           else KERNEL_BY_TYPE[args.type]),
                "kernel_ms": round(k_mean, 4),
                "kernel_ms_min_median_max": [round(float(kernel_ms.min()), 4), round(float(np.median(kernel_ms)), 4),
                                             round(float(kernel_ms.max()), 4)],
                "kernel_launches_timed": int(kernel_ms.size),
                "shader_mhz": shader_mhz,
                "box_probe": box_probe,  # torch's fill / copy over the output buffer: GB/s written, GB/s read + written
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "cpu_baseline": cpu,
        }
        result = json.dumps(line)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (printed last: RCCL writes a line of its own to stdout when it is first used)
        sys.stdout.flush()
        print(result, flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Decode-throughput benchmark of the MI355X DINT path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N > 1: starts one process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the decode kernel over the rank's whole resident
collection (every unit of every posting list). Inputs (encoded stream, unit
table, dictionary) are in HBM before the timed region starts. Posting lists
are partitioned statically across ranks, the dictionary is replicated, and
there is no data-path collective: the only collectives are the dictionary
broadcast during set-up and the max-over-ranks of the elapsed time.

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
    "gov2": dict(universe=25_000_000, postings=1.0e9, replicate=5),
    "clueweb": dict(universe=50_000_000, postings=1.25e9, replicate=1),
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
                    help="gov2: universe 25M, 1e9 postings encoded per GPU, decoded x5 from distinct addresses (5e9 integers "
                         "per step: SURVEY §8d config 2); clueweb: universe 50M, 1.25e9 postings per GPU (config 4 at 8 GPUs)")
    ap.add_argument("--postings", type=float, default=None, help="postings encoded per GPU (weak scaling: fixed per GPU)")
    ap.add_argument("--replicate", type=int, default=None,
                    help="device-side copies of the encoded shard at distinct addresses, all decoded in one step")
    ap.add_argument("--universe", type=int, default=None, help="documents")
    ap.add_argument("--unit-ints", type=int, default=8192)
    ap.add_argument("--dict-sample", type=float, default=2.0e7,
                    help="postings the DSF dictionary statistics are collected from")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="decode time budget of each leg of the CPU baseline sample (0 = skip)")
    ap.add_argument("--traffic-file", default=None,
                    help="JSON written by tools/pmc_traffic.py from rocprofv3 --pmc passes of THIS command; without it "
                         "roofline.traffic is null (HBM counters cannot be read from inside the run)")
    ap.add_argument("--no-verify", action="store_true", help="skip the full bit-exact output check")
    return ap.parse_args(argv)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` on its own: start N fresh processes, one per GPU, and relay rank 0's line.
    (This process has not touched the GPU; the children are new processes, not a re-exec.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(kind, dict_file, enc, list_byte_starts, seconds):
    """The oracle's restatement of the reference decode, timed on this host: (i) one thread, per-list timing
    summed exactly like vroom_env/decode.cpp:139-150; (ii) every core, the lists statically partitioned by
    stream bytes, wall time. A bounded sample: each leg stops after about `seconds` of decode time."""
    import threading

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
    cores = len(os.sched_getaffinity(0))
    # thread k takes the lists that begin in [k, k + 1) / cores of the stream's bytes (a thread may get none)
    cut = np.searchsorted(list_byte_starts, [enc.size * k // cores for k in range(1, cores)])
    bounds = [0] + [int(list_byte_starts[i]) if i < len(list_byte_starts) else enc.size for i in cut] + [enc.size]
    res = [None] * cores

    def work(k, t0):
        a, b = bounds[k], bounds[k + 1]
        n = passes_k = 0
        while b > a and time.perf_counter() - t0 < seconds:  # whole passes over the thread's range
            n += od.time_stream(enc[a:b])[1]
            passes_k += 1
        res[k] = (n, passes_k)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k, t0)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    tot = sum(r[0] for r in res)
    one["all_cores"] = {"value": round(tot / wall / 1e6, 2), "unit": "M ints/s", "cores": cores,
                        "sample": f"{cores} threads, contiguous list ranges of equal stream bytes, whole passes until {seconds:.0f}s: "
                                  f"{tot} postings in {wall:.1f}s wall (first thread start to last thread end)"}
    return one


def main():
    args = parse_args()
    w = WORKLOADS[args.workload]
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

    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))

    kind = host.KIND_BY_TYPE[args.type]
    threads = max(1, host.default_threads() // world)

    # ---- set-up (untimed): collection shard, dictionary, encode, upload ----------
    t0 = time.time()
    p = host.synth_params(universe=universe, seed=args.seed)
    # The collection is `world * postings` postings; list lengths are drawn once
    # (same on every rank) and contiguous list ranges balanced by postings are
    # handed to the ranks.
    lens_all = host.synth_lengths(p, postings * world)
    lo, hi = sharding.partition_lists(lens_all, world)[rank]
    lens = lens_all[lo:hi]
    gaps = host.synth_gaps(p, lens, first_list_id=lo, threads=threads)
    coll = host.Collection(gaps, lens)
    log(rank, f"rank shard: lists [{lo},{hi}) = {coll.num_postings} postings, generated in {time.time() - t0:.1f}s")

    t0 = time.time()
    if rank == 0:
        # dictionary statistics from a prefix sample of the collection (rank 0's first lists)
        dict_file = host.build_dictionary(kind, coll, max_sample_ints=int(args.dict_sample), threads=threads)
    else:
        dict_file = None
    if distributed:
        box = [dict_file]
        dist.broadcast_object_list(box, src=0)
        dict_file = box[0]
    log(rank, f"dictionary: {len(dict_file)} B in {time.time() - t0:.1f}s")

    t0 = time.time()
    enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=args.unit_ints, threads=threads)
    bpi = enc.size * 8 / coll.num_postings
    log(rank, f"encoded: {enc.size} B ({bpi:.3f} bits/int), {len(units)} units in {time.time() - t0:.1f}s")

    d = device.Dictionary(kind, dict_file, device=local_rank)
    info = d.info()
    n_ints = coll.num_postings * R
    enc_dev = torch.empty(enc.size * R, dtype=torch.uint8, device=dev)
    enc_one = torch.from_numpy(enc).to(dev)
    units_all = np.tile(units, R)
    for r in range(R):
        enc_dev[r * enc.size:(r + 1) * enc.size].copy_(enc_one)
        sl = slice(r * len(units), (r + 1) * len(units))
        units_all["in_off"][sl] += np.uint64(r * enc.size)
        units_all["out_off"][sl] += np.uint64(r * coll.num_postings)
    del enc_one
    units_dev = device.units_to_device(units_all, dev)
    n_units = len(units_all)
    out_dev = torch.empty(n_ints, dtype=torch.int32, device=dev)
    end_dev = torch.zeros(n_units, dtype=torch.int64, device=dev)
    log(rank, f"device buffers: enc {enc_dev.data_ptr():#x} ({enc_dev.data_ptr() % (2 << 20):#x} mod 2 MB) "
              f"out {out_dev.data_ptr():#x} units {units_dev.data_ptr():#x}")

    def sync_all():
        if distributed:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    # ---- warm-up ------------------------------------------------------------------
    for _ in range(args.warmup):
        d.decode_units(enc_dev, units_dev, n_units, out_dev, end_dev)
    sync_all()

    # ---- timed region: exactly K steps (each returns the end offsets too, like the reference's decode) ----
    t_start = time.perf_counter()
    for _ in range(args.steps):
        d.decode_units(enc_dev, units_dev, n_units, out_dev, end_dev)
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
    bit_exact = None
    if not args.no_verify:
        bit_exact = True
        for r in range(R):  # one replica at a time: the output is 4 bytes x 5e9 at the default size
            got = out_dev[r * coll.num_postings:(r + 1) * coll.num_postings].cpu().numpy().view(np.uint32)
            bit_exact = bit_exact and bool(np.array_equal(got, coll.gaps))
            del got
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
        st = d.stream_stats(enc)
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
            prev_end = np.r_[np.uint64(0), ends[: len(units)][np.flatnonzero(np.r_[first[1:], True])][:-1]]
            cpu = cpu_baseline(kind, dict_file, enc, prev_end.astype(np.int64), args.cpu_seconds)

        algo_bytes = 4 * n_ints + payload_bytes  # per launch, this rank
        traffic, traffic_note = None, "not measured in this process (rocprofv3 --pmc passes: tools/profile_round.sh)"
        if args.traffic_file and os.path.exists(args.traffic_file):
            with open(args.traffic_file) as f:
                tf = json.load(f)
            if tf.get("type") == args.type and tf.get("ints_per_launch") == n_ints:
                traffic = {"write_gb": tf["write_gb"], "fetch_gb_raw": tf["fetch_gb_raw"],
                           "fetch_gb_corrected": tf["fetch_gb_corrected"],
                           "total_gb": round(tf["write_gb"] + tf["fetch_gb_corrected"], 3)}
                traffic_note = tf.get("note", "")
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
            "dtype": "u32",
            "data": "synthetic",
            "bit_exact": bit_exact,
            "config": {
                "workload": f"{args.type} decode, DSF-65536-16 dictionary (hot set in LDS), {args.workload}-shaped synthetic "
                            f"docIDs: universe {universe}, {postings} postings encoded per GPU"
                            + (f", decoded x{R} per step from {R} device-side copies at distinct addresses" if R > 1 else ""),
                "ints_per_gpu_per_step": n_ints,
                "lists_per_gpu": int(np.count_nonzero(lens)) * R,
                "units_per_gpu": n_units,
                "unit_ints": args.unit_ints,
                "bits_per_int": round(bpi, 3),
                **stream,
                "hot_codewords_in_lds": int(info.hot_entries),
                "lds_bytes": int(info.lds_bytes),
                "parallelism": f"list-range x{world}",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_note": traffic_note,
                "kernel": KERNEL_BY_TYPE[args.type],
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
        print(json.dumps(line), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

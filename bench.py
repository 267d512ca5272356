#!/usr/bin/env python3
"""Decode-throughput benchmark of the MI355X DINT path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
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
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)  # (the shader clock is still ramping for the first few launches)
    ap.add_argument("--type", default="single_packed_dint",
                    choices=["single_rect_dint", "single_packed_dint", "multi_packed_dint"])
    ap.add_argument("--postings", type=float, default=1.0e9,
                    help="postings encoded per GPU (weak scaling: fixed per GPU)")
    ap.add_argument("--replicate", type=int, default=1,
                    help="device-side copies of the encoded shard at distinct addresses (scale knob)")
    ap.add_argument("--universe", type=int, default=25_000_000, help="documents (Gov2-shaped: 25M)")
    ap.add_argument("--unit-ints", type=int, default=8192)
    ap.add_argument("--dict-sample", type=float, default=2.0e7,
                    help="postings the DSF dictionary statistics are collected from")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="summed decode time budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-verify", action="store_true", help="skip the full bit-exact output check")
    return ap.parse_args()


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")

    from dint_amd import device, host, sharding

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    kind = host.KIND_BY_TYPE[args.type]
    postings = int(args.postings)
    threads = max(1, host.default_threads() // world)

    # ---- set-up (untimed): collection shard, dictionary, encode, upload ----------
    t0 = time.time()
    params = dict(universe=args.universe, seed=args.seed)
    p = host.synth_params(**params)
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
    R = max(1, args.replicate)
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

    log(rank, f"device buffers: enc {enc_dev.data_ptr():#x} out {out_dev.data_ptr():#x} units {units_dev.data_ptr():#x}")

    def sync_all():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- warm-up ------------------------------------------------------------------
    for _ in range(args.warmup):
        d.decode_units(enc_dev, units_dev, n_units, out_dev, end_dev)
    sync_all()

    # ---- timed region: exactly K steps ----------------------------------------------
    t_start = time.perf_counter()
    for _ in range(args.steps):
        d.decode_units(enc_dev, units_dev, n_units, out_dev, None)
    sync_all()
    elapsed = time.perf_counter() - t_start
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_ints], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_ints = int(tot.item())
    else:
        total_ints = n_ints

    # ---- per-launch kernel time (HIP events on the launch stream) -------------------
    kernel_ms = []
    for _ in range(max(3, min(args.steps, 10))):
        d.decode_units(enc_dev, units_dev, n_units, out_dev, None)
        torch.cuda.synchronize(dev)
        kernel_ms.append(d.last_kernel_ms())
    kernel_ms_avg = float(np.mean(kernel_ms))

    # ---- correctness: bit-exact against the encoder's input -------------------------
    ends = end_dev.cpu().numpy().view(np.uint64)
    payload_bytes = int((ends - units_all["in_off"]).sum())
    bit_exact = None
    if not args.no_verify:
        got = out_dev.cpu().numpy().view(np.uint32)
        bit_exact = all(np.array_equal(got[r * coll.num_postings:(r + 1) * coll.num_postings], coll.gaps)
                        for r in range(R))
        if distributed:
            ok = torch.tensor([1 if bit_exact else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            bit_exact = bool(ok.item())
        if not bit_exact:
            raise SystemExit("FATAL: decoded integers differ from the encoder's input")
        del got

    # ---- CPU baseline (rank 0, N=1 only): the oracle timed like vroom_env/decode.cpp ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle  # the CPU restatement, used here as the timed baseline only

        od = oracle.OracleDict(kind, dict_file)
        # whole passes over the stream (or a prefix of it, if one pass is longer than the budget)
        # until about cpu_seconds of decode time have been summed
        sec = ints = lists = passes = 0
        while sec < args.cpu_seconds:
            s1, i1, l1 = od.time_stream(enc, max_seconds=args.cpu_seconds - sec)
            sec, ints, lists, passes = sec + s1, ints + i1, lists + l1, passes + 1
        cpu = {
            "value": round(ints / sec / 1e6, 2), "unit": "M ints/s", "cores": 1, "kind": "port",
            "sample": f"{passes} pass(es) over the same encoded stream, {lists} list decodes ({ints} postings), "
                      f"per-list timing summed as in vroom_env/decode.cpp:139-150, {sec:.1f}s of decode time",
        }

    if rank == 0:
        algo_bytes = 4 * n_ints + payload_bytes  # per launch, this rank
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
        # this same command; tools/pmc_traffic.py stores bytes per decoded integer under profiles/.
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                t = json.load(f)
            if t.get("type") == args.type and t.get("postings_per_gpu") == n_ints:
                traffic = round(t["hbm_bytes_per_launch"] / 1e9, 3)
        achieved = algo_bytes / (kernel_ms_avg * 1e-3) / 1e9
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
                "workload": f"{args.type} decode, DSF-65536-16 dictionary (hot set in LDS), Gov2-shaped synthetic "
                            f"docIDs: universe {args.universe}, {postings} postings/GPU"
                            + (f" x{R} device-side replicas" if R > 1 else ""),
                "postings_per_gpu": n_ints,
                "lists_per_gpu": int(np.count_nonzero(lens)) * R,
                "units_per_gpu": n_units,
                "unit_ints": args.unit_ints,
                "bits_per_int": round(bpi, 3),
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
                "traffic_unit": "GB per launch (FETCH_SIZE + WRITE_SIZE, profiles/traffic.json)",
                "kernel": "decode_single_kernel",
                "kernel_ms": round(kernel_ms_avg, 4),
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

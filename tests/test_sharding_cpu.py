"""List-range sharding and the world_size-2 path of bench.py's partition logic on gloo."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from dint_amd import host, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_partition_covers_every_list_once(world):
    r = np.random.default_rng(world)
    lens = (r.pareto(1.2, 5000) * 20 + 1).astype(np.uint32)
    parts = sharding.partition_lists(lens, world)
    assert parts[0][0] == 0 and parts[-1][1] == len(lens)
    assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    loads = [int(lens[a:b].sum()) for a, b in parts]
    assert sum(loads) == int(lens.sum())
    assert max(loads) - min(loads) <= 2 * int(lens.max())     # balanced by postings, not by lists


def test_partition_units_rebases_offsets(small_corpus):
    enc, units = small_corpus.encoded(host.SINGLE_PACKED)
    parts = sharding.partition_units(units, 3)
    assert sum(len(p[0]) for p in parts) == len(units)
    pos = 0
    for part, byte_lo, byte_hi, int_lo, int_hi in parts:
        if not len(part):
            continue
        assert int_lo == pos and part["out_off"][0] == 0
        pos = int_hi
        lists = np.unique(part["list"])
        assert lists.min() > -1
    assert pos == small_corpus.coll.num_postings


WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "oracle")]
    import numpy as np, torch, torch.distributed as dist
    from dint_amd import host, sharding
    import oracle
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    p = host.synth_params(universe=300000, seed=99)
    lens_all = host.synth_lengths(p, 200000 * world)
    lo, hi = sharding.partition_lists(lens_all, world)[rank]
    lens = lens_all[lo:hi]
    coll = host.Collection(host.synth_gaps(p, lens, first_list_id=lo, threads=2), lens)
    box = [host.build_dictionary(host.SINGLE_PACKED, coll, threads=2) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)                 # dictionary replicated (set-up, not data path)
    enc, units = host.encode_vroom(host.SINGLE_PACKED, box[0], coll, unit_ints=2048, threads=2)
    out, _ = oracle.OracleDict(oracle.SINGLE_PACKED, box[0]).decode_stream(enc, coll.num_postings)
    assert np.array_equal(out, coll.gaps)
    # the only reductions of the multi-GPU path: total ints (sum), elapsed (max), checksum (sum)
    t = torch.tensor([coll.num_postings, int(out.astype(np.uint64).sum() % (1 << 62))], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    el = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    # every rank regenerates the whole collection to check the shards tile it exactly
    whole = host.synth_gaps(p, lens_all, first_list_id=0, threads=2)
    assert t[0].item() == whole.size and t[1].item() == int(whole.astype(np.uint64).sum() % (1 << 62)) % (1 << 63)
    assert el.item() == float(world)
    dist.destroy_process_group()
    print("rank", rank, "ok", coll.num_postings)
""")


def test_two_ranks_on_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert all("ok" in o for o in outs)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no torchrun) starts two fresh processes, which rendezvous, shard the lists
    and reduce — on CPU over gloo, with the device layer stubbed at the test level (tests/bench_stub.py)."""
    import json

    env = dict(os.environ, DINT_BENCH_STUB="bench_stub", PYTHONPATH=os.path.join(ROOT, "tests"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--postings", "200000", "--replicate", "2", "--universe", "300000", "--dict-sample", "100000",
                        "--cpu-seconds", "0", "--no-verify"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "list-range x2"
    assert line["value"] > 0 and line["roofline"]["traffic"] is None
    assert line["data"] == "stub" and line["stub"] == "bench_stub"  # a stubbed run can not pass for a measurement
    assert line["config"]["process_group"] == "gloo"


def _stub_bench(extra, env_extra, timeout=300):
    env = dict(os.environ, DINT_BENCH_STUB="bench_stub", PYTHONPATH=os.path.join(ROOT, "tests"), **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--postings", "200000",
                           "--universe", "300000", "--dict-sample", "100000", "--cpu-seconds", "0", "--no-verify"] + extra,
                          env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("how", ["BENCH_STUB_FAIL_RANK", "BENCH_STUB_HANG_RANK"])
def test_bench_launcher_stops_every_rank_when_one_fails_or_hangs(how):
    """A rank that dies (or never answers) must not leave its siblings waiting in a collective and the caller waiting
    for them: the launcher stops all of them and exits non-zero, without a result line."""
    import time

    t0 = time.time()
    r = _stub_bench(["--gpus", "2", "--rank-timeout", "20"], {how: "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "all ranks stopped" in r.stderr
    assert time.time() - t0 < 120


def test_bench_as_rank_takes_that_ranks_shard():
    """--as-rank 3/8: one process decoding the shard rank 3 of an 8-rank job would get."""
    import json

    r = _stub_bench(["--gpus", "1", "--as-rank", "3/8"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    p = host.synth_params(universe=300000, seed=12345)
    lens_all = host.synth_lengths(p, 200000 * 8)
    lo, hi = sharding.partition_lists(lens_all, 8)[3]
    assert line["emulated_rank"] == "3/8" and f"lists [{lo},{hi})" in line["emulated_note"]
    assert line["config"]["ints_per_gpu_per_step"] == int(lens_all[lo:hi].sum()) * 5  # (gov2 workload: x5 replicas)
    assert line["config"]["parallelism"] == "list-range shard 3 of 8" and line["n_gpus"] == 1
    bad = _stub_bench(["--gpus", "1", "--as-rank", "8/8"], {})
    assert bad.returncode != 0


def test_one_rank_and_two_ranks_decode_the_same_workload_per_gpu():
    """--multi-rank-distinct on: a two-rank job generates, encodes and decodes per GPU what a one-GPU run does — the two
    lines' config.workload differ in the rank count alone (and the collection's total, which is the count times the
    per-GPU postings), `replicate` is 1 in both and the per-GPU distinct postings agree within a list's length. `off`:
    a fifth... here, the multi-rank variant of the workload — x5 replicas — and the line says so at its top level."""
    import json
    import re

    def line_of(extra):
        r = _stub_bench(extra, {})
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])

    one = line_of(["--gpus", "1"])
    two = line_of(["--gpus", "2", "--multi-rank-distinct", "on"])
    off = line_of(["--gpus", "2", "--multi-rank-distinct", "off"])
    norm = lambda w: re.sub(r"\d+ GPU\(s\): contiguous list ranges of one collection of \d+ postings", "<ranks>", w)
    assert one["config"]["workload"] != two["config"]["workload"]
    assert norm(one["config"]["workload"]) == norm(two["config"]["workload"])
    assert "1 GPU(s)" in one["config"]["workload"] and "2 GPU(s)" in two["config"]["workload"]
    assert one["replicate"] == two["replicate"] == 1 and off["replicate"] == 5
    assert abs(one["distinct_postings_per_gpu"] - two["distinct_postings_per_gpu"]) < 100_000
    assert one["config"]["ints_per_gpu_per_step"] == one["distinct_postings_per_gpu"]
    assert off["config"]["ints_per_gpu_per_step"] == 5 * off["distinct_postings_per_gpu"]
    assert "decoded x5" in off["config"]["workload"] and "decoded x" not in two["config"]["workload"]


def test_bench_help_renders():
    """argparse formats every help string with %: an unescaped per cent sign in one of them breaks --help."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "--placement-trials" in r.stdout


def test_bench_at_world_size_8_through_its_own_launcher():
    """The dress rehearsal of `python bench.py --gpus 8` without a node: eight ranks started by bench.py's own launcher,
    rendezvous on 127.0.0.1 over gloo, the eight list-range shards of ONE collection (every list in exactly one shard,
    the shards' postings within a list's length of each other), the dictionary broadcast from rank 0, the reductions —
    with the stubbed device layer. The line reports the whole job: 8 x the per-GPU integers per step."""
    import json

    env = dict(os.environ, DINT_BENCH_STUB="bench_stub", PYTHONPATH=os.path.join(ROOT, "tests"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--postings", "100000", "--universe", "300000",
                        "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-verify", "--dict-sample", "50000",
                        "--rank-timeout", "240"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 8 and line["data"] == "stub" and line["scaling"] == "weak"
    assert line["config"]["process_group"] == "gloo" and line["config"]["parallelism"] == "list-range x8"
    per_gpu = line["config"]["ints_per_gpu_per_step"]
    # (gov2 with more than one rank: the shard decoded x5 from five copies — here rank 0's share of 8 x 100000 postings)
    assert per_gpu % 5 == 0 and abs(per_gpu // 5 - 100000) < 100000 // 3
    assert line["value"] > 0 and line["steps"] == 2
    from dint_amd import host, sharding

    lens = host.synth_lengths(host.synth_params(universe=300000, seed=12345), 100000 * 8)
    parts = sharding.partition_lists(lens, 8)
    assert parts[0][0] == 0 and parts[-1][1] == len(lens) and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    sizes = [int(lens[a:b].sum()) for a, b in parts]
    assert sum(sizes) == int(lens.sum()) and max(sizes) - min(sizes) <= 2 * int(lens.max())
    assert per_gpu == 5 * sizes[0]

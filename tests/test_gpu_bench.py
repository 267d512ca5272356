"""bench.py end to end on one GPU at a small size: the driver's command form, the line's contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--type", "multi_packed_dint", "--unit-ints", "256"], ["--type", "single_rect_dint", "--workload", "clueweb"]],
                         ids=["single_packed", "multi_block_units", "single_rect_clueweb"])
def test_bench_line_contract(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--postings", "3e6",
           "--replicate", "2", "--dict-sample", "1e6", "--cpu-seconds", "0.5"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["bit_exact"] is True and d["value"] > 0 and d["higher_is_better"] is True and d["dtype"] == "u32"
    assert "workload" in d["config"] and "model" not in d["config"]
    # by default the pair of big buffers is chosen among six candidates each (the library's dint_unit_table_rank_outputs);
    # the line says so, carries every candidate's kernel time and what the process's first allocation reached
    trials = d["config"]["placement_trial_kernel_ms"]
    assert "candidate output buffers" in d["config"]["placement"] and len(trials["output_buffers"]) == 6 and len(trials["stream_buffers"]) == 6
    assert d["roofline"]["kernel_ms_first_allocation"] == trials["output_buffers"][0] and d["value_first_allocation"] > 0
    assert d["config"]["ints_per_gpu_per_step"] >= 6_000_000 * 0.9 and d["config"]["distinct_postings_per_gpu"] * 2 == d["config"]["ints_per_gpu_per_step"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel_launches_timed"] == 3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["kernel_ms"] > 0
    lo, med, hi = rf["kernel_ms_min_median_max"]
    assert lo <= med <= hi
    assert rf["traffic"] is None  # not measured in this process: only a same-command PMC file may fill it
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["all_cores"]["cores"] >= 1 and cb["cpu_model"]


def test_bench_placement_trials_are_in_the_line():
    """--placement-trials N: candidate output buffers, then candidate stream buffers, decoded into during set-up; the line
    carries every candidate's kernel time, which pair was kept, and what the FIRST allocation reached. --placement-trials 1:
    the first allocation and nothing else."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--postings", "3e6",
           "--chunk-postings", "1e6", "--dict-sample", "1e6", "--cpu-seconds", "0", "--placement-trials", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    trials = d["config"]["placement_trial_kernel_ms"]
    assert "candidate output buffers" in d["config"]["placement"] and len(trials["output_buffers"]) == 3
    assert len(trials["stream_buffers"]) == 3 and min(trials["stream_buffers"]) <= min(trials["output_buffers"])
    assert d["roofline"]["kernel_ms_first_allocation"] == trials["output_buffers"][0] and d["value_first_allocation"] > 0
    assert d["bit_exact"] is True and d["config"]["distinct_postings_per_gpu"] == d["config"]["ints_per_gpu_per_step"]  # three pieces, one copy
    r = subprocess.run(cmd[:-1] + ["1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["placement"].startswith("first allocation") and d["config"]["placement_trial_kernel_ms"] is None
    assert d["value_first_allocation"] == d["value"] and d["roofline"]["frac_first_allocation"] == d["roofline"]["frac"]


def test_rank_outputs_is_a_library_call(tmp_path):
    """dint_unit_table_rank_outputs: candidate output buffers for a prepared unit table, each decoded, timed and left holding
    the decoded integers; the fastest is named."""
    import numpy as np
    import torch
    from dint_amd import device, host

    coll = host.synth_collection(2_000_000, universe=2_000_000, seed=5)
    kind = host.SINGLE_PACKED
    dict_file = host.build_dictionary(kind, coll, max_sample_ints=1_000_000)
    enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=4096)
    dev = torch.device("cuda", 0)
    d = device.Dictionary(kind, dict_file)
    enc_dev = torch.from_numpy(enc).to(dev)
    units_dev = device.units_to_device(units, dev)
    table = device.UnitTable(d, enc_dev, units_dev, len(units), coll.num_postings)
    outs = [torch.full((coll.num_postings,), -1, dtype=torch.int32, device=dev) for _ in range(3)]
    ms, best = table.rank_outputs(outs)
    assert len(ms) == 3 and all(m > 0 for m in ms) and ms[best] == min(ms)
    for o in outs:
        assert np.array_equal(o.cpu().numpy().view(np.uint32), coll.gaps)
    with pytest.raises(device.DintError):
        table.rank_outputs([torch.empty(coll.num_postings - 1, dtype=torch.int32, device=dev)])
    table.close()


@pytest.mark.parametrize("k", [0, 3, 7])
def test_config4_shard_of_eight_on_one_gpu(k):
    """BASELINE config 4: single_packed_dint, ClueWeb09-shaped lists partitioned over 8 ranks by postings — the shard rank k
    would get (sharding.partition_lists(lens_all, 8)[k], dictionary from rank 0's sample), decoded by one process on one
    GPU at reduced postings and verified bit-exact inside bench.py."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "clueweb",
           "--type", "single_packed_dint", "--postings", "3e7", "--dict-sample", "1e6", "--cpu-seconds", "0", "--as-rank", f"{k}/8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["bit_exact"] is True and d["emulated_rank"] == f"{k}/8" and d["n_gpus"] == 1
    assert d["config"]["parallelism"] == f"list-range shard {k} of 8"
    # the shard holds its share of the 8 x 3e7 postings (balanced by postings: within one longest list, universe / 3)
    assert abs(d["config"]["ints_per_gpu_per_step"] - 30_000_000) < 50_000_000 / 3
    assert "universe 50000000" in d["config"]["workload"]


def test_rccl_process_group_initialises_on_hardware():
    """The N > 1 path's collectives (RCCL all-reduce of elapsed / integers / ok, the dictionary broadcast) with a world of
    one: init_process_group("nccl") and every reduction run on the GPU at least once."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--postings", "2e6",
           "--dict-sample", "1e6", "--cpu-seconds", "0", "--force-process-group"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(key, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["process_group"] == "nccl" and d["bit_exact"] is True and d["n_gpus"] == 1

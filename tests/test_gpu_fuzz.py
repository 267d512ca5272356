"""Differential fuzzing on the GPU: the HIP decode path (C ABI) against the generator's substitution — the same cases
tests/test_fuzz_cpu.py pins against the C oracle and the naive Python decoder — random dictionary files for the three
formats x random decoder-legal slot streams NO ENCODER EMITS (vroom_env/dint_codecs.hpp:45-100, :536-612;
include/dint/dint_codecs.hpp:21-46 accept any slot sequence): 28 800 vroom lists cut into random units, 6 000 posting
lists (full blocks of random slots, short blocks interpolative with and without the stored sum), AND queries over them.
Integers, docIDs, freqs AND end offsets, bit for bit."""
import json
import os

import numpy as np
import pytest

import fuzz_streams as F
import oracle
from queries import intersect, intersect_freqs

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_digests.json")))
VROOM = F.plan(*GOLDEN["vroom_plan"])
INDEX = F.index_plan(*GOLDEN["index_plan"])
assert all(str(c[0]) in GOLDEN["vroom"] for c in VROOM) and all(str(c[0]) in GOLDEN["index"] for c in INDEX)


@pytest.fixture(scope="module")
def device():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from dint_amd import device as dev  # fails loudly if libdint_hip.so is missing

    return dev


@pytest.fixture(autouse=True)
def _options_back_to_default(device):
    yield
    device.reset_options()


def _decode_with_canary(device, d, enc, units, total):
    """decode_units into a buffer with a canary behind the last integer and in front of the first."""
    import torch

    dev = torch.device("cuda", 0)
    enc_dev = torch.from_numpy(np.ascontiguousarray(enc)).to(dev)
    out_dev = torch.full((total + 128,), -1, dtype=torch.int32, device=dev)
    end_dev = torch.zeros(max(1, len(units)), dtype=torch.int64, device=dev)
    d.decode_units(enc_dev, device.units_to_device(units, dev), len(units), out_dev[64:], end_dev)
    torch.cuda.synchronize()
    got = out_dev.cpu().numpy().view(np.uint32)
    assert (got[:64] == 0xFFFFFFFF).all() and (got[64 + total:] == 0xFFFFFFFF).all(), "wrote outside [0, n)"
    return got[64:64 + total], end_dev.cpu().numpy().view(np.uint64)[:len(units)]


@pytest.mark.parametrize("case", VROOM, ids=lambda c: f"seed{c[0]}")
def test_vroom_case(device, case):
    D, S = F.build_case(case)
    pinned = GOLDEN["vroom"].get(str(case[0]))  # (tests/fuzz_soak.py runs this function over seeds of its own: nothing pinned)
    assert pinned is None or F.digest(D, S) == pinned["digest"]
    d = device.Dictionary(D.kind, D.file)
    total = len(S.expect)
    # the generator's own units: cut at random slot (multi: block) boundaries, odd addresses, lists of one integer
    out, ends = _decode_with_canary(device, d, S.enc, S.units, total)
    bad = np.flatnonzero(out != S.expect)
    assert bad.size == 0, f"first difference at integer {bad[0]}: {out[bad[0]]} != {S.expect[bad[0]]}"
    assert np.array_equal(ends, S.ends)
    # the host pre-pass over the same bytes (dint_index_stream: the framing loop of vroom_env/decode.cpp:139-150):
    # whole lists, block-sized units, odd-sized units
    for unit_ints in (0, 256, 77):
        units, n_ints, n_lists = d.index_stream(S.enc, unit_ints)
        assert (n_ints, n_lists) == (total, len(S.lists))
        if unit_ints == 0:
            assert np.array_equal(units["in_off"], [l[0] for l in S.lists])
            assert np.array_equal(units["n"], [l[1] for l in S.lists])
        out, ends = _decode_with_canary(device, d, S.enc, units, total)
        assert np.array_equal(out, S.expect)
        last = np.r_[units["list"][1:] != units["list"][:-1], True]
        assert np.array_equal(ends[last], S.ends[np.r_[S.units["list"][1:] != S.units["list"][:-1], True]])
    # a PREPARED table of block-granular units (multi-dictionary streams: the units that fit no tile are cut in two records
    # each when the table is made, split_units_kernel — and, the option off, decoded by the general kernel)
    if D.kind == F.MULTI:
        import torch

        dev = torch.device("cuda", 0)
        units, _, _ = d.index_stream(S.enc, 256)
        enc_dev = torch.from_numpy(np.ascontiguousarray(S.enc)).to(dev)
        units_dev = device.units_to_device(units, dev)
        want_ends = None
        for split in (1, 0):
            device.set_option("split_units", split)
            table = device.UnitTable(d, enc_dev, units_dev, len(units), total)
            out_dev = torch.full((total + 64,), -1, dtype=torch.int32, device=dev)
            end_dev = torch.zeros(len(units), dtype=torch.int64, device=dev)
            for _ in range(2):
                table.decode(out_dev[:total], end_dev)
            torch.cuda.synchronize()
            got = out_dev.cpu().numpy().view(np.uint32)
            assert np.array_equal(got[:total], S.expect) and (got[total:] == 0xFFFFFFFF).all(), split
            ends = end_dev.cpu().numpy().view(np.uint64)
            if want_ends is None:
                want_ends = ends.copy()
            assert np.array_equal(ends, want_ends)
            table.close()
        last = np.r_[units["list"][1:] != units["list"][:-1], True]
        assert np.array_equal(want_ends[last], S.ends[np.r_[S.units["list"][1:] != S.units["list"][:-1], True]])
        # ... and of the generator's own units, several blocks each: the table finds the blocks once (refine_units_kernel)
        # and decodes a table of blocks — or, the option off, the units as they came
        own_dev = device.units_to_device(S.units, dev)
        for refine in (1, 0):
            device.set_option("refine_units", refine)
            table = device.UnitTable(d, enc_dev, own_dev, len(S.units), total)
            out_dev = torch.full((total + 64,), -1, dtype=torch.int32, device=dev)
            end_dev = torch.zeros(len(S.units), dtype=torch.int64, device=dev)
            table.decode(out_dev[:total])  # (no end offsets asked for)
            table.decode(out_dev[:total], end_dev)
            torch.cuda.synchronize()
            got = out_dev.cpu().numpy().view(np.uint32)
            assert np.array_equal(got[:total], S.expect) and (got[total:] == 0xFFFFFFFF).all(), refine
            assert np.array_equal(end_dev.cpu().numpy().view(np.uint64), S.ends), refine
            table.close()
    # the one-list call (Coder::decode's shape) on a sample
    for off, n, first in S.lists[::40]:
        got, used = d.decode_list(S.enc, off, n)
        assert np.array_equal(got, S.expect[first:first + n])


def _fuzz_queries(r, X, n_queries):
    lens = np.diff(X.bounds)
    n_lists = len(lens)
    qs = []
    for _ in range(n_queries):
        k = int(r.integers(1, 5))
        qs.append(r.integers(0, n_lists, k).astype(np.uint32))
    big = np.argsort(-lens, kind="stable")[:12]
    for _ in range(n_queries // 2):
        qs.append(r.choice(big, int(r.integers(2, 4))).astype(np.uint32))
    return qs


@pytest.mark.parametrize("case", INDEX, ids=lambda c: f"seed{c[0]}")
def test_index_case(device, case):
    import torch

    Dd, Df, X = F.build_index_case(case)
    pinned = GOLDEN["index"].get(str(case[0]))
    assert pinned is None or F.index_digest(Dd, Df, X) == pinned["digest"]
    dd, fd = device.Dictionary(Dd.kind, Dd.file), device.Dictionary(Df.kind, Df.file)
    blocks, total = device.index_posting_lists(X.index, X.offsets)
    assert total == len(X.docids) and int(blocks["n"].sum()) == total
    # the one-shot call
    docids, freqs = device.decode_posting_lists(dd, fd, X.index, blocks, total)
    assert np.array_equal(docids, X.docids) and np.array_equal(freqs, X.freqs)
    # a prepared table, taught and untaught, with and without freqs
    dev = torch.device("cuda", 0)
    padded = np.concatenate([X.index, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    taught, plain = device.BlockTable(dd, blocks, padded.size), device.BlockTable(dd, blocks, padded.size)
    taught.learn(dd, fd, index_dev, padded.size)
    for table, passes in ((taught, 1), (plain, 3)):
        for p in range(passes):
            docids_dev = torch.full((total + 64,), -1, dtype=torch.int32, device=dev)
            freqs_dev = torch.full((total + 64,), -1, dtype=torch.int32, device=dev)
            with_freqs = not (table is plain and p == 0)
            table.decode(dd, fd if with_freqs else None, index_dev, padded.size, docids_dev[:total],
                         freqs_dev[:total] if with_freqs else None)
            torch.cuda.synchronize()
            got = docids_dev.cpu().numpy().view(np.uint32)
            assert np.array_equal(got[:total], X.docids) and (got[total:] == 0xFFFFFFFF).all()
            got = freqs_dev.cpu().numpy().view(np.uint32)
            assert (got[total:] == 0xFFFFFFFF).all()
            if with_freqs:
                assert np.array_equal(got[:total], X.freqs)
    # AND queries over the fuzzed lists: counts against plain set intersection and the oracle's and_query
    r = np.random.default_rng(case[0])
    qs = _fuzz_queries(r, X, 120)
    qi = device.QueryIndex(dd, X.index, X.offsets)
    want = np.array([intersect(X.docids, X.bounds, q) for q in qs], dtype=np.uint64)
    assert np.array_equal(qi.and_queries(qs), want)
    for q, w in list(zip(qs, want))[::15]:
        assert int(qi.and_queries([q])[0]) == int(w)
    oi = oracle.OracleIndex(oracle.OracleDict(Dd.kind, Dd.file), X.index, X.offsets, int(X.docids.max()) + 1)
    of = oracle.OracleDict(Df.kind, Df.file)
    counts, sums, _ = qi.and_queries_with_freqs(fd, qs)
    assert np.array_equal(counts, want)
    for i in range(0, len(qs), 9):
        n, fsum, _ = oi.and_query_freqs(of, qs[i])
        assert (n, fsum) == (int(counts[i]), int(sums[i])) and n == oi.and_query(qs[i])
    qi.close()


def test_a_multi_file_with_small_contexts_fills_the_lds_image(device):
    """choose_hot_set (hip_dictionary.inc): what the dictionaries that fit whole leave of their even quota goes to the
    others — three reserved-only contexts beside three full ones must not leave half the image empty — and the hot / cold
    border, now at a different codeword in every context, decodes like the generator says."""
    r = np.random.default_rng(60606)
    shape = dict(m_entries=65536, value_profile=("mixed", "tiny", "mixed"), size_profile="pow2",
                 context_entries=[7, 65536, 7, 65536, 7, 40000])
    D = F.make_dictionary(r, F.MULTI, **shape)
    S = F.make_stream(r, D, 300)
    full = F.make_dictionary(np.random.default_rng(60607), F.SINGLE, m_entries=65536, value_profile="mixed", size_profile="pow2")
    d, s = device.Dictionary(D.kind, D.file), device.Dictionary(full.kind, full.file)
    assert d.info().lds_bytes >= 0.95 * s.info().lds_bytes, (d.info().lds_bytes, s.info().lds_bytes)
    out, ends = _decode_with_canary(device, d, S.enc, S.units, len(S.expect))
    assert np.array_equal(out, S.expect) and np.array_equal(ends, S.ends)
    st = d.stream_stats(S.enc)
    assert 0 < st.hot_codewords < st.codewords  # (both sides of the border are exercised)

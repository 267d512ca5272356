"""GPU parity: the HIP decode path (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import oracle
from dint_amd import host

pytestmark = pytest.mark.gpu

SINGLE_KINDS = [host.SINGLE_PACKED, host.RECTANGULAR]
ALL_KINDS = SINGLE_KINDS + [host.MULTI_PACKED]


@pytest.fixture(scope="module")
def device():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from dint_amd import device as dev  # fails loudly if libdint_hip.so is missing

    return dev


def test_wave_scan(device):
    r = np.random.default_rng(0)
    for _ in range(8):
        v = r.integers(0, 300, 64).astype(np.uint32)
        assert np.array_equal(device.debug_wave_scan(v), np.cumsum(v, dtype=np.uint32))
    assert np.array_equal(device.debug_wave_scan(np.ones(64)), np.arange(1, 65, dtype=np.uint32))


@pytest.mark.parametrize("kind", ALL_KINDS)
@pytest.mark.parametrize("corpus_name", ["small_corpus", "dense_corpus", "sparse_corpus"])
def test_stream_matches_oracle(device, request, kind, corpus_name):
    corpus = request.getfixturevalue(corpus_name)
    enc, units = corpus.encoded(kind)
    d = device.Dictionary(kind, corpus.dict_file(kind))
    out, ends, _ = device.decode_stream(d, enc, units, corpus.coll.num_postings)
    want, _ = oracle.OracleDict(kind, corpus.dict_file(kind)).decode_stream(enc, corpus.coll.num_postings)
    assert np.array_equal(want, corpus.coll.gaps)          # the oracle agrees with the encoder's input
    assert np.array_equal(out, want)                       # the device agrees with the oracle
    # end offsets: each unit ends where the next unit of the same list starts
    same = units["list"][1:] == units["list"][:-1]
    assert np.array_equal(ends[:-1][same], units["in_off"][1:][same])
    assert ends[-1] == enc.size


@pytest.mark.parametrize("kind", ALL_KINDS)
@pytest.mark.parametrize("greedy", [False, True])
def test_oracle_encoded_stream_decodes_on_the_device(device, small_corpus, kind, greedy):
    """the stream the ORACLE's encoder restatement writes (oracle/dint_oracle_encode.c:
    vroom_env/dint_codecs.hpp:110-518 + jobs.hpp:74-95 over the corpus as a .docs file) — not a byte of it from the
    product's encoder — indexed by the product's host pre-pass and decoded by the HIP kernels: == the gaps."""
    if greedy and kind == host.MULTI_PACKED:
        pytest.skip("the reference has no greedy multi-dictionary coder")
    coll, dict_file = small_corpus.coll, small_corpus.dict_file(kind)
    ids = host.gaps_to_docids(coll)
    b = coll.list_bounds()
    words = host.collection_words([ids[int(b[i]):int(b[i + 1])] for i in range(len(coll.lens))], num_docs=int(ids.max()) + 1)
    enc, lists, ints = oracle.OracleBuilder(kind, dict_file).encode_collection(words, docs=True, greedy=greedy)
    assert ints == coll.num_postings
    d = device.Dictionary(kind, dict_file)
    for unit_ints in (256, 4096):
        units, total, n_lists = d.index_stream(enc, unit_ints)
        assert (total, n_lists) == (ints, lists)
        out, ends, _ = device.decode_stream(d, enc, units, total)
        assert np.array_equal(out, coll.gaps)
        assert int(ends[-1]) == enc.size


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_index_stream_equals_encoder_sidecar(device, small_corpus, kind):
    """The host pre-pass finds the same list boundaries the encoder recorded."""
    enc, units = small_corpus.encoded(kind)
    d = device.Dictionary(kind, small_corpus.dict_file(kind))
    for unit_ints in (0, 256, 4096):
        idx, total, lists = d.index_stream(enc, unit_ints)
        assert total == small_corpus.coll.num_postings
        assert lists == np.count_nonzero(small_corpus.coll.lens)
        assert int(idx["n"].sum()) == total
        out, _, _ = device.decode_stream(d, enc, idx, total)
        assert np.array_equal(out, small_corpus.coll.gaps)
    first = np.r_[True, units["list"][1:] != units["list"][:-1]]
    idx0, _, _ = d.index_stream(enc, 0)
    assert np.array_equal(idx0["in_off"], units["in_off"][first])


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_decode_list_call_shape(device, small_corpus, kind):
    """Coder::decode(dict, in, out, universe, n) -> in_end, one list at a time."""
    enc, _ = small_corpus.encoded(kind)
    od = oracle.OracleDict(kind, small_corpus.dict_file(kind))
    d = device.Dictionary(kind, small_corpus.dict_file(kind))
    off = 0
    checked = 0
    while off < enc.size and checked < 40:
        n, _, payload = oracle.header_read(enc, off)
        want, used = od.decode_list(enc, payload, n)
        got, consumed = d.decode_list(enc, payload, n)
        assert np.array_equal(got, want)
        assert consumed == used
        off = payload + used
        checked += 1


# ---- hand-assembled known-answer vectors through the C ABI -------------------------------------
from kat import DICT_FILES, cases  # noqa: E402


@pytest.mark.parametrize("kind", SINGLE_KINDS)
@pytest.mark.parametrize("case", cases("single_cases"), ids=lambda c: c[0])
def test_single_kat(device, kind, case):
    name, buf, off, n, expect = case
    d = device.Dictionary(kind, DICT_FILES[kind])
    got, consumed = d.decode_list(buf, off, n)
    assert np.array_equal(got, expect)
    assert consumed == buf.size - off


@pytest.mark.parametrize("kind", SINGLE_KINDS)
def test_all_kats_in_one_launch(device, kind):
    """Every vector as one unit of a single batched decode: units at arbitrary byte offsets, outputs
    back to back, a canary after the last integer."""
    import torch

    cs = cases("single_cases")
    blob, units, pos = [], [], 0
    from dint_amd.host import UNIT_DTYPE

    out_pos = 0
    for i, (name, buf, off, n, expect) in enumerate(cs):
        blob.append(buf)
        units.append((pos + off, out_pos, n, i))
        pos += buf.size
        out_pos += n
    enc = np.concatenate(blob + [np.zeros(16, dtype=np.uint8)])
    table = np.array(units, dtype=UNIT_DTYPE)
    d = device.Dictionary(kind, DICT_FILES[kind])
    dev = torch.device("cuda", 0)
    enc_dev = torch.from_numpy(enc).to(dev)
    out_dev = torch.full((out_pos + 64,), -1, dtype=torch.int32, device=dev)
    end_dev = torch.zeros(len(table), dtype=torch.int64, device=dev)
    d.decode_units(enc_dev, device.units_to_device(table, dev), len(table), out_dev, end_dev)
    torch.cuda.synchronize()
    got = out_dev.cpu().numpy().view(np.uint32)
    assert np.array_equal(got[:out_pos], np.concatenate([c[4] for c in cs]))
    assert (got[out_pos:] == 0xFFFFFFFF).all()              # nothing written past the last integer
    ends = end_dev.cpu().numpy()
    assert np.array_equal(ends, np.cumsum([c[1].size for c in cs]))


@pytest.mark.parametrize("unit_ints", [64, 255, 1000, 100_000])
def test_any_unit_size_gives_the_same_integers(device, small_corpus, unit_ints):
    d = device.Dictionary(host.SINGLE_PACKED, small_corpus.dict_file(host.SINGLE_PACKED))
    enc, _ = small_corpus.encoded(host.SINGLE_PACKED)
    units, total, _ = d.index_stream(enc, unit_ints)
    out, _, _ = device.decode_stream(d, enc, units, total)
    assert np.array_equal(out, small_corpus.coll.gaps)


@pytest.mark.parametrize("kind,unit_ints", [(host.SINGLE_PACKED, 8192), (host.RECTANGULAR, 8192), (host.MULTI_PACKED, 8192),
                                            (host.MULTI_PACKED, 256)],
                         ids=["single_packed", "single_rect", "multi_packed", "multi_packed_block_units"])
def test_full_size_properties(device, kind, unit_ints):
    """4e7 postings, well past the sizes the oracle is used on: bit-exact against the encoder's input, the
    units' end offsets are where the encoder put the next unit, and decode is idempotent (a second pass over
    the same buffer changes nothing). unit_ints = 256 on a multi-dictionary stream: block-granular units
    (vroom_env/dint_codecs.hpp:521-619 decodes the blocks one after the other; with the block table they are
    independent)."""
    import torch

    coll = host.synth_collection(40_000_000, universe=25_000_000, seed=2024)
    dict_file = host.build_dictionary(kind, coll, max_sample_ints=5_000_000)
    enc, units = host.encode_vroom(kind, dict_file, coll, unit_ints=unit_ints)
    if unit_ints == 256:
        assert int(units["n"].max()) <= 256 and len(units) > coll.num_postings // 300
    d = device.Dictionary(kind, dict_file)
    dev = torch.device("cuda", 0)
    enc_dev = torch.from_numpy(enc).to(dev)
    units_dev = device.units_to_device(units, dev)
    out_dev = torch.zeros(coll.num_postings, dtype=torch.int32, device=dev)
    end_dev = torch.zeros(len(units), dtype=torch.int64, device=dev)
    d.decode_units(enc_dev, units_dev, len(units), out_dev, end_dev)
    first = out_dev.clone()
    d.decode_units(enc_dev, units_dev, len(units), out_dev)
    torch.cuda.synchronize()
    assert torch.equal(first, out_dev)
    got = out_dev.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, coll.gaps)
    assert int(got.sum(dtype=np.uint64)) == int(coll.gaps.sum(dtype=np.uint64))
    # a unit ends where the next one of its list begins (lists are separated by their vroom headers)
    ends = end_dev.cpu().numpy().astype(np.uint64)
    nxt = units["in_off"][1:].astype(np.uint64)
    same_list = units["list"][1:] == units["list"][:-1]
    assert np.array_equal(ends[:-1][same_list], nxt[same_list])
    assert (ends[:-1][~same_list] <= nxt[~same_list]).all() and int(ends[-1]) <= enc.size


def test_malformed_unit_tables_stay_inside_their_own_output(device, small_corpus):
    """Unit tables come from the caller: entries that point past the stream, claim more integers than the stream
    holds, or lie outside the output are skipped or decode garbage INSIDE THEIR OWN [out_off, out_off + n) — every
    other unit is exact, nothing is written elsewhere, nothing faults."""
    import torch

    kind = host.SINGLE_PACKED
    d = device.Dictionary(kind, small_corpus.dict_file(kind))
    enc, units = small_corpus.encoded(kind)
    total = small_corpus.coll.num_postings
    units = units.copy()
    n_units = len(units)
    assert n_units > 64
    capacity = total + 1500      # room behind the last unit for one that claims too much
    canaries = 4096
    bad = {}
    bad["in_off_past_end"] = 5
    units["in_off"][5] = enc.size + 100
    bad["in_off_near_end"] = 9
    units["in_off"][9] = enc.size - 3
    bad["in_off_huge"] = 13
    units["in_off"][13] = 2**63 + 17
    bad["n_past_capacity"] = 21
    units["n"][21] = capacity            # out_off + n > capacity: skipped
    bad["out_off_past_capacity"] = 33
    units["out_off"][33] = capacity + 5
    bad["n_zero"] = 40
    units["n"][40] = 0
    bad["out_off_wraps"] = 55
    units["out_off"][55] = 2**64 - 8     # out_off + n wraps around to a small number
    bad["n_over_limit"] = 47
    units["n"][47] = device.MAX_UNIT_INTS + 1
    last = n_units - 1                   # claims 1000 integers more than its stream holds: garbage, but its own
    units["n"][last] += 1000
    dev = torch.device("cuda", 0)
    enc_dev = torch.from_numpy(enc).to(dev)
    units_dev = device.units_to_device(units, dev)
    sentinel = -1412567295  # 0xABCDEF01
    out_full = torch.full((capacity + canaries,), sentinel, dtype=torch.int32, device=dev)
    for _ in range(2):
        d.decode_units(enc_dev, units_dev, n_units, out_full[:capacity])
    torch.cuda.synchronize()
    got = out_full.cpu().numpy()
    assert (got[capacity:] == sentinel).all(), "written past the output's capacity"
    good = small_corpus.encoded(kind)[1]
    want = small_corpus.coll.gaps.view(np.int32)
    touched = set(bad.values()) | {last}
    for u in range(n_units):
        lo, n = int(good["out_off"][u]), int(good["n"][u])
        if u not in touched:
            assert np.array_equal(got[lo:lo + n], want[lo:lo + n]), f"unit {u} is not exact"
    for name in ("in_off_past_end", "in_off_huge", "n_past_capacity", "out_off_past_capacity", "n_zero", "n_over_limit", "out_off_wraps"):
        u = bad[name]
        lo, n = int(good["out_off"][u]), int(good["n"][u])
        assert (got[lo:lo + n] == sentinel).all() or name in ("in_off_past_end", "in_off_huge"), name
    # the last unit: its real integers are exact, the rest of what it claims is its own to fill
    lo, n = int(good["out_off"][last]), int(good["n"][last])
    assert np.array_equal(got[lo:lo + n], want[lo:lo + n])
    assert (got[lo + n + 1000:capacity] == sentinel).all()


def test_units_over_the_size_limit_are_rejected(device, small_corpus):
    kind = host.SINGLE_PACKED
    d = device.Dictionary(kind, small_corpus.dict_file(kind))
    enc, _ = small_corpus.encoded(kind)
    with pytest.raises(device.DintError):
        d.decode_list(enc, 0, device.MAX_UNIT_INTS + 1)
    units, total, _ = d.index_stream(enc, 1 << 31)   # asks for units larger than the limit: capped, still exact
    assert int(units["n"].max()) <= device.MAX_UNIT_INTS
    out, _, _ = device.decode_stream(d, enc, units, total)
    assert np.array_equal(out, small_corpus.coll.gaps)


@pytest.mark.parametrize("case", cases("multi_cases"), ids=lambda c: c[0])
def test_multi_kat(device, case):
    name, buf, off, n, expect = case
    d = device.Dictionary(host.MULTI_PACKED, DICT_FILES[2])
    got, consumed = d.decode_list(buf, off, n)
    assert np.array_equal(got, expect)
    assert consumed == buf.size - off


def test_multi_blocks_use_both_codeword_widths(small_corpus):
    """Guard against a vacuous multi test: the corpus must exercise 16-bit AND 8-bit blocks."""
    enc, units = small_corpus.encoded(host.MULTI_PACKED)
    sels = enc[units["in_off"].astype(np.int64)]
    assert (sels < 6).any() and (sels >= 6).any() and (sels < 12).all()


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.RECTANGULAR])
@pytest.mark.parametrize("unit_ints", [1024, 16])
def test_bundles_of_tiny_lists(device, kind, unit_ints):
    """Collections made of tiny lists (the long tail of an inverted index): consecutive tiny units are
    decoded several to a tile (decode_bundle). Same integers and end offsets as one unit at a time."""
    import os

    for seed, params in ((5, dict(max_len=70)), (6, dict(max_len=9)), (8, dict(max_len=300, universe=3_000_000))):
        coll = host.synth_collection(60_000, seed=seed, **{"universe": 200_000, **params})
        dict_file = host.build_dictionary(kind, coll)
        enc, _ = host.encode_vroom(kind, dict_file, coll, unit_ints=unit_ints)
        d = device.Dictionary(kind, dict_file)
        units, total, _ = d.index_stream(enc, unit_ints)
        out, ends, _ = device.decode_stream(d, enc, units, total)
        assert np.array_equal(out, coll.gaps)
        with device.options(bundles=0):
            out1, ends1, _ = device.decode_stream(d, enc, units, total)
        assert np.array_equal(out1, coll.gaps)
        assert np.array_equal(ends, ends1)
        # a unit ends where the next one starts, minus that list's header when it opens a new list
        same_list = units["list"][1:] == units["list"][:-1]
        assert np.array_equal(ends[:-1][same_list], units["in_off"][1:][same_list])
        assert int(ends[-1]) == enc.size


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.RECTANGULAR, host.MULTI_PACKED])
def test_seeded_sweep_of_small_collections(device, kind):
    """Many small collections with different length / density mixes, unit sizes from one block to whole
    lists: decoded integers and end offsets must equal the encoder's input and its byte layout."""
    r = np.random.default_rng(1234 + kind)
    for it in range(24):
        postings = int(r.integers(2_000, 120_000))
        universe = int(r.choice([5_000, 60_000, 1_000_000, 25_000_000]))
        max_len = int(r.choice([8, 70, 300, 5_000, universe // 3 + 1]))
        unit_ints = int(r.choice([0, 256, 300, 1024, 8192]))
        coll = host.synth_collection(postings, seed=int(r.integers(1 << 30)), universe=universe, max_len=max(1, max_len))
        dict_file = host.build_dictionary(kind, coll)
        enc, _ = host.encode_vroom(kind, dict_file, coll, unit_ints=unit_ints, greedy=bool(it % 5 == 0))
        d = device.Dictionary(kind, dict_file)
        units, total, n_lists = d.index_stream(enc, unit_ints)
        assert total == coll.num_postings and n_lists == np.count_nonzero(coll.lens)
        out, ends, _ = device.decode_stream(d, enc, units, total)
        assert np.array_equal(out, coll.gaps), (it, postings, universe, max_len, unit_ints)
        same_list = units["list"][1:] == units["list"][:-1]
        assert np.array_equal(ends[:-1][same_list], units["in_off"][1:][same_list])
        assert int(ends[-1]) == enc.size


@pytest.mark.parametrize("kind", ALL_KINDS)
@pytest.mark.parametrize("corpus_name", ["small_corpus", "sparse_corpus"])
def test_block_statistics_on_the_device(device, request, kind, corpus_name):
    """SURVEY §8 f2 against the ORACLE (oracle/dint_oracle_stats.c: selector::get, adjusted::collect, the saving
    filter, freq_length_sorter and the DSF cut restated from statistics_collectors.hpp:21-118, block_statistics.hpp:82-106,
    dictionary_builders.hpp:15-75): the device's n-gram counts (dint_count_ngrams) are the oracle's, n-gram for n-gram
    and context for context, and the device's selection (dint_select_ngrams) is the oracle's, in the same order — the
    order among entries of equal frequency and length is by their integers in both (the reference's depends on
    libstdc++: oracle header). The dictionary file built from it equals the host library's."""
    import torch

    coll = request.getfixturevalue(corpus_name).coll
    multi = kind == host.MULTI_PACKED
    gaps = np.ascontiguousarray(coll.gaps, dtype=np.uint32)
    st = oracle.Stats(multi, gaps)
    st.collect_lists(coll.lens)
    gaps_dev = torch.from_numpy(gaps.view(np.int32)).cuda()
    starts = np.zeros(len(coll.lens) + 1, dtype=np.uint64)
    np.cumsum(coll.lens, out=starts[1:])
    entries, ms = device.count_ngrams(gaps_dev, starts, multi)
    ngram = lambda e: tuple(int(x) for x in gaps[int(e["pos"]):int(e["pos"]) + int(e["len"])])
    got = {(int(e["ctx"]), ngram(e)): int(e["freq"]) for e in entries}
    want = {(c, st.ngram(e)): int(e["freq"]) for c in range(st.contexts) for e in st.entries(c)}
    assert len(got) == len(entries) and got == want
    chosen = device.select_ngrams(gaps_dev, entries, coll.num_postings, top_k=65536)
    want_sel = [(c, int(e["freq"]), st.ngram(e)) for c in range(st.contexts) for e in st.select(c)[0]]
    assert [(int(e["ctx"]), int(e["freq"]), ngram(e)) for e in chosen] == want_sel
    built, ms2 = device.build_dictionary(kind, coll)
    assert built == host.build_dictionary(kind, coll)
    sampled, _ = device.build_dictionary(kind, coll, max_sample_ints=coll.num_postings // 3)
    assert sampled == host.build_dictionary(kind, coll, max_sample_ints=coll.num_postings // 3)
    assert ms > 0 and ms2 > 0


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_selection_on_the_device_is_in_dictionary_order(device, small_corpus, kind):
    """dint_select_ngrams: every context's kept n-grams most frequent first, then the longer, then by their integers; at
    most 65536 per context; nothing the filter drops; and exactly the entries the host's selection appends."""
    import torch

    coll = small_corpus.coll
    gaps = np.ascontiguousarray(coll.gaps, dtype=np.uint32)
    gaps_dev = torch.from_numpy(gaps.view(np.int32)).cuda()
    starts = np.zeros(len(coll.lens) + 1, dtype=np.uint64)
    np.cumsum(coll.lens, out=starts[1:])
    entries, _ = device.count_ngrams(gaps_dev, starts, kind == host.MULTI_PACKED)
    chosen = device.select_ngrams(gaps_dev, entries, coll.num_postings, top_k=65536)
    small = device.select_ngrams(gaps_dev, entries, coll.num_postings, top_k=100)
    assert 0 < len(chosen) <= len(entries)
    key = lambda e: (int(e["ctx"]), -int(e["freq"]), -int(e["len"]), tuple(int(x) for x in gaps[int(e["pos"]):int(e["pos"]) + int(e["len"])]))
    keys = [key(e) for e in chosen]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    for c in np.unique(chosen["ctx"]):
        assert int((chosen["ctx"] == c).sum()) <= 65536
        mine = [k for k in keys if k[0] == int(c)]
        assert [key(e) for e in small if int(e["ctx"]) == int(c)] == mine[:100]
    assert host.pack_dictionary(kind, gaps, chosen) == host.build_dictionary(kind, coll)


def test_block_context_wraps_like_the_reference(device):
    """selector::get adds its 1 in uint32_t (statistics_collectors.hpp:23,36): a block holding 0xFFFFFFFF is context 0;
    0xFFFFFFFE is context 5. The device's counting puts the blocks' n-grams where the oracle's does."""
    import torch

    gaps = np.full(512, 3, dtype=np.uint32)
    gaps[7] = 0xFFFFFFFF
    gaps[256 + 9] = 0xFFFFFFFE
    starts = np.array([0, 512], dtype=np.uint64)
    entries, _ = device.count_ngrams(torch.from_numpy(gaps.view(np.int32)).cuda(), starts, multi=True)
    st = oracle.Stats(True, gaps)
    st.collect(0, 512)
    ngram = lambda e: tuple(int(x) for x in gaps[int(e["pos"]):int(e["pos"]) + int(e["len"])])
    got = {(int(e["ctx"]), ngram(e)): int(e["freq"]) for e in entries}
    want = {(c, st.ngram(e)): int(e["freq"]) for c in range(st.contexts) for e in st.entries(c)}
    assert got == want
    assert {c for c, _ in got} == {0, 5} and got[(0, (0xFFFFFFFF,))] == 1 and got[(5, (0xFFFFFFFE,))] == 1


def test_ngram_counts_of_a_tiny_collection(device):
    import torch

    gaps = np.array([5, 5, 5, 5, 1, 2, 1, 2, 9], dtype=np.uint32)           # two lists: 8 + 1 integers
    starts = np.array([0, 8, 9], dtype=np.uint64)
    entries, _ = device.count_ngrams(torch.from_numpy(gaps.view(np.int32)).cuda(), starts, multi=False)
    got = {(tuple(gaps[int(e["pos"]):int(e["pos"]) + int(e["len"])])): int(e["freq"]) for e in entries}
    want = {(5, 5, 5, 5, 1, 2, 1, 2): 1, (5, 5, 5, 5): 1, (1, 2, 1, 2): 1, (5, 5): 2, (1, 2): 2, (5,): 4, (1,): 2, (2,): 2, (9,): 1}
    assert got == want
    none, _ = device.count_ngrams(torch.from_numpy(gaps.view(np.int32)).cuda(), starts, multi=True)   # no whole 256-block
    assert none.size == 0


@pytest.mark.parametrize("kind,unit_ints", [(host.SINGLE_PACKED, 2048), (host.RECTANGULAR, 64), (host.MULTI_PACKED, 256),
                                            (host.MULTI_PACKED, 4096)])
def test_prepared_unit_table_decodes_like_decode_units(device, small_corpus, kind, unit_ints):
    """dint_unit_table_create + dint_decode_unit_table: the bundle schedule built once, then one launch per decode —
    the same integers and end offsets as dint_decode_units, launch after launch, on a side stream too; a smaller
    output than the table was prepared for is refused."""
    import torch

    d = device.Dictionary(kind, small_corpus.dict_file(kind))
    enc, _ = small_corpus.encoded(kind)
    units, total, _ = d.index_stream(enc, unit_ints)
    dev = torch.device("cuda", 0)
    enc_dev = torch.from_numpy(enc).to(dev)
    units_dev = device.units_to_device(units, dev)
    want, want_ends, _ = device.decode_stream(d, enc, units, total)
    assert np.array_equal(want, small_corpus.coll.gaps)
    table = device.UnitTable(d, enc_dev, units_dev, len(units), total)
    side = torch.cuda.Stream(dev)
    for rep in range(3):
        out_dev = torch.full((total + 512,), -7, dtype=torch.int32, device=dev)
        end_dev = torch.zeros(len(units), dtype=torch.int64, device=dev)
        if rep == 1:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                table.decode(out_dev, end_dev, stream=side.cuda_stream)
            side.synchronize()
        else:
            table.decode(out_dev, end_dev)
            torch.cuda.synchronize()
        got = out_dev.cpu().numpy()
        assert np.array_equal(got[:total].view(np.uint32), want)
        assert (got[total:] == -7).all()
        assert np.array_equal(end_dev.cpu().numpy().view(np.uint64), want_ends)
    small = torch.empty(total - 1, dtype=torch.int32, device=dev)
    with pytest.raises(device.DintError):
        table.decode(small)
    table.close()


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_two_shards_two_handles_two_streams(device, kind):
    """The closest a one-GPU box gets to the multi-GPU layout (one dint_dict handle per device): the two
    list-range shards of partition_lists(lens, 2) of ONE collection, each with a dictionary handle of its own (the
    replicated dictionary, built from shard 0's lists like bench.py does) and a prepared unit table, decoded at the same
    time on two streams of device 0, three times; every shard bit-exact, and together they are the collection."""
    import torch
    from dint_amd import sharding

    p = host.synth_params(universe=2_000_000, seed=77)
    lens_all = host.synth_lengths(p, 5_000_000)
    parts = sharding.partition_lists(lens_all, 2)
    colls = [host.Collection(host.synth_gaps(p, lens_all[lo:hi], first_list_id=lo), lens_all[lo:hi]) for lo, hi in parts]
    assert sum(c.num_postings for c in colls) == int(lens_all.sum())
    dict_file = host.build_dictionary(kind, colls[0], max_sample_ints=1_000_000)
    dev = torch.device("cuda", 0)
    jobs = []
    for c in colls:
        d = device.Dictionary(kind, dict_file)  # a handle per shard: its own queue slots, events, schedule workspaces
        enc, units = host.encode_vroom(kind, dict_file, c, unit_ints=256 if kind == host.MULTI_PACKED else 4096)
        enc_dev = torch.from_numpy(enc).to(dev)
        units_dev = device.units_to_device(units, dev)
        table = device.UnitTable(d, enc_dev, units_dev, len(units), c.num_postings)
        jobs.append(dict(d=d, c=c, enc_dev=enc_dev, units_dev=units_dev, table=table, stream=torch.cuda.Stream(dev),
                         out=torch.empty(c.num_postings, dtype=torch.int32, device=dev), n_units=len(units)))
    torch.cuda.synchronize(dev)
    for rep in range(3):
        for j in jobs:
            j["out"].fill_(-1)
        torch.cuda.synchronize(dev)
        for j in jobs:  # both launches are in flight together
            with torch.cuda.stream(j["stream"]):
                j["table"].decode(j["out"], None, stream=j["stream"].cuda_stream)
        for j in jobs:
            j["stream"].synchronize()
            assert np.array_equal(j["out"].cpu().numpy().view(np.uint32), j["c"].gaps), rep
    whole = host.synth_gaps(p, lens_all, first_list_id=0)
    assert np.array_equal(np.concatenate([j["out"].cpu().numpy().view(np.uint32) for j in jobs]), whole)
    for j in jobs:
        j["table"].close()


@pytest.mark.parametrize("split", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("kind", [host.MULTI_PACKED, host.SINGLE_PACKED])
def test_every_ticket_size_decodes_the_same(device, small_corpus, kind, split):
    """The bundle path hands out 1/2^n of a 64-unit chunk per ticket (32 counters, contiguous ranges, stealing); n follows the
    launch's size unless dint_set_option(chunk_split) says otherwise. Every n decodes a block-granular multi-dictionary
    stream (bundles and nothing else) and an in-index table (three decodes: general kernel, then the kernels without the
    unit queue) to the same integers."""
    import torch
    from test_index_cpu import get_index

    dev = torch.device("cuda", 0)
    with device.options(chunk_split=split):
        if kind == host.MULTI_PACKED:
            dict_file = small_corpus.dict_file(kind)
            enc, units = host.encode_vroom(kind, dict_file, small_corpus.coll, unit_ints=256)
            d = device.Dictionary(kind, dict_file)
            enc_dev = torch.from_numpy(enc).to(dev)
            units_dev = device.units_to_device(units, dev)
            table = device.UnitTable(d, enc_dev, units_dev, len(units), small_corpus.coll.num_postings)
            for _ in range(2):
                out_dev = torch.full((small_corpus.coll.num_postings + 64,), -1, dtype=torch.int32, device=dev)
                table.decode(out_dev[:small_corpus.coll.num_postings], None)
                torch.cuda.synchronize()
                got = out_dev.cpu().numpy()
                assert np.array_equal(got[:-64].view(np.uint32), small_corpus.coll.gaps) and (got[-64:] == -1).all()
            table.close()
        ix = get_index(small_corpus, kind)
        blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
        dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
        padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
        index_dev = torch.from_numpy(padded).to(dev)
        bt = device.BlockTable(dd, blocks, padded.size)
        for i in range(4):
            docids_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
            freqs_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
            bt.decode(dd, fd, index_dev, padded.size, docids_dev, freqs_dev)
            torch.cuda.synchronize()
            assert np.array_equal(docids_dev.cpu().numpy().view(np.uint32), ix.docids), i
            assert np.array_equal(freqs_dev.cpu().numpy().view(np.uint32), ix.freqs), i
        bt.close()


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_a_stream_that_straddles_4_gib(device, small_corpus, kind):
    """dint_unit::in_off is 64-bit: the same stream placed across offset 2^32 of a 4.3 GB device buffer decodes to the same
    integers and end offsets (units of 256: the bundle paths; units of 4096: the segment path)."""
    import torch

    dev = torch.device("cuda", 0)
    d = device.Dictionary(kind, small_corpus.dict_file(kind))
    enc, _ = small_corpus.encoded(kind)
    for unit_ints in (256, 4096):
        units, total, _ = d.index_stream(enc, unit_ints)
        want, want_ends, _ = device.decode_stream(d, enc, units, total)
        assert np.array_equal(want, small_corpus.coll.gaps)
        for cut in (enc.size // 2, 4099):
            shift = (1 << 32) - cut
            big = torch.zeros(shift + enc.size + 16, dtype=torch.uint8, device=dev)
            big[shift:shift + enc.size] = torch.from_numpy(enc).to(dev)
            moved = units.copy()
            moved["in_off"] += np.uint64(shift)
            out_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
            end_dev = torch.zeros(len(units), dtype=torch.int64, device=dev)
            units_dev = device.units_to_device(moved, dev)
            d.decode_units(big, units_dev, len(moved), out_dev, end_dev)
            torch.cuda.synchronize()
            assert np.array_equal(out_dev.cpu().numpy().view(np.uint32), want), (unit_ints, cut)
            assert np.array_equal(end_dev.cpu().numpy().view(np.uint64), want_ends + np.uint64(shift)), (unit_ints, cut)
            table = device.UnitTable(d, big, units_dev, len(moved), total)
            out_dev.fill_(-1)
            table.decode(out_dev, end_dev)
            torch.cuda.synchronize()
            assert np.array_equal(out_dev.cpu().numpy().view(np.uint32), want), (unit_ints, cut, "table")
            table.close()
            del big
            torch.cuda.empty_cache()


@pytest.mark.parametrize("corpus_name", ["small_corpus", "sparse_corpus"])
def test_units_that_fit_no_tile_are_cut_in_two(device, request, corpus_name):
    """A prepared multi-dictionary table of block-granular units: the blocks of more than 63 lanes of stream (8-bit blocks with
    many literals — the sparse corpus is full of them) are cut at a codeword boundary into two bundle records each when the
    table is made (split_units_kernel) and decoded by a second launch of the bundles kernel; with the option off the general
    kernel decodes them. Same integers, same end offsets, nothing written past the last integer."""
    import torch

    corpus = request.getfixturevalue(corpus_name)
    kind = host.MULTI_PACKED
    enc, _ = corpus.encoded(kind)
    d = device.Dictionary(kind, corpus.dict_file(kind))
    units, total, _ = d.index_stream(enc, 256)
    dev = torch.device("cuda", 0)
    enc_dev = torch.from_numpy(enc).to(dev)
    units_dev = device.units_to_device(units, dev)
    results = []
    try:
        for split in (1, 0):
            device.set_option("split_units", split)
            table = device.UnitTable(d, enc_dev, units_dev, len(units), total)
            out_dev = torch.full((total + 64,), -1, dtype=torch.int32, device=dev)
            end_dev = torch.zeros(len(units), dtype=torch.int64, device=dev)
            for _ in range(2):
                table.decode(out_dev[:total], end_dev)
            torch.cuda.synchronize()
            got = out_dev.cpu().numpy().view(np.uint32)
            assert np.array_equal(got[:total], corpus.coll.gaps) and (got[total:] == 0xFFFFFFFF).all()
            results.append(end_dev.cpu().numpy().view(np.uint64).copy())
            table.close()
    finally:
        device.reset_options()
    assert np.array_equal(results[0], results[1])
    # (how many units the walk had to cut: the spans of the stream say)
    nxt = np.r_[units["in_off"][1:], enc.size].astype(np.int64)
    span = nxt - units["in_off"].astype(np.int64) - 1
    narrow = enc[units["in_off"].astype(np.int64)] >= 6
    assert int(np.count_nonzero((narrow & (span > 252)) | (~narrow & (span > 504)))) > (0 if corpus_name == "small_corpus" else 20)

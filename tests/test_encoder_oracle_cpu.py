"""the product's ENCODERS against the oracle's line-cited restatement (oracle/dint_oracle_encode.c) —
"product bytes == oracle bytes" for the whole-list coders (single_opt_dint, single_greedy_dint, multi_opt_dint:
vroom_env/dint_codecs.hpp:110-518), the block coders and interpolative tails (include/dint/dint_codecs.hpp:52-458,
block_codecs.hpp:104-128), dict_posting_list::write (dict_posting_list.hpp:10-56), the vroom framing over a collection
file (encode.cpp:133-191, jobs.hpp:74-95, binary_collection.hpp:131-146), the hash-only lookup quirks of SURVEY H9, the
packing (dictionary_building_utils.hpp:241-292) and the three ingest tools on written .docs / .freqs files."""
import json
import os
import subprocess

import numpy as np
import pytest

import kat
import oracle
from dint_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dint_amd", "bin")
KINDS = [host.RECTANGULAR, host.SINGLE_PACKED, host.MULTI_PACKED]
TYPE_OF = {host.RECTANGULAR: "single_rect_dint", host.SINGLE_PACKED: "single_packed_dint", host.MULTI_PACKED: "multi_packed_dint"}


def corpus_words(coll, docs=True):
    """The corpus as a collection file's words: .docs (docIDs behind the leading `1, num_docs`) or .freqs (gaps + 1)."""
    b = coll.list_bounds()
    if docs:
        ids = host.gaps_to_docids(coll)
        lists = [ids[int(b[i]):int(b[i + 1])] for i in range(len(coll.lens))]
        return host.collection_words(lists, num_docs=int(ids.max()) + 1)
    lists = [coll.gaps[int(b[i]):int(b[i + 1])] + np.uint32(1) for i in range(len(coll.lens))]
    return host.collection_words(lists)


@pytest.mark.parametrize("corpus_name", ["small_corpus", "dense_corpus", "sparse_corpus"])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("greedy", [False, True])
def test_vroom_stream_bytes_equal_the_oracles(request, corpus_name, kind, greedy):
    """every list's header and payload: product encoder (parallel, cut list, zero-run table) == oracle (the reference's
    node path walked back through the dummy nodes), for all three kinds x opt / greedy x the three corpora; and the
    oracle's stream decodes (oracle decoder) to the input."""
    if greedy and kind == host.MULTI_PACKED:
        pytest.skip("the reference has no greedy multi-dictionary coder")
    corpus = request.getfixturevalue(corpus_name)
    coll, dict_file = corpus.coll, corpus.dict_file(kind)
    enc, _units = host.encode_vroom(kind, dict_file, coll, unit_ints=1024, greedy=greedy)
    builder = oracle.OracleBuilder(kind, dict_file)
    words = corpus_words(coll, docs=True)
    want, lists, ints = builder.encode_collection(words, docs=True, greedy=greedy)
    assert lists == int((coll.lens > 0).sum()) and ints == coll.num_postings
    assert enc.size == want.size and np.array_equal(enc, want)
    # the product's own collection reader + framing gives the same bytes again
    enc2, units2, lists2, ints2 = host.encode_collection(kind, dict_file, words, docs=True, unit_ints=1024, greedy=greedy)
    assert (lists2, ints2) == (lists, ints) and np.array_equal(enc2, want)
    assert int(units2["n"].sum()) == ints
    got, _ = oracle.OracleDict(kind, dict_file).decode_stream(want, coll.num_postings)
    assert np.array_equal(got, coll.gaps)


@pytest.mark.parametrize("kind", KINDS)
def test_freqs_file_framing(small_corpus, kind):
    """a .freqs file: no leading singleton, every value minus one, universe = u32 sum (jobs.hpp:74-84)."""
    coll, dict_file = small_corpus.coll, small_corpus.dict_file(kind)
    words = corpus_words(coll, docs=False)
    want, lists, ints = oracle.OracleBuilder(kind, dict_file).encode_collection(words, docs=False)
    enc, _u, lists2, ints2 = host.encode_collection(kind, dict_file, words, docs=False)
    assert (lists, ints) == (lists2, ints2) == (int((coll.lens > 0).sum()), coll.num_postings)
    assert np.array_equal(enc, want)
    enc3, _ = host.encode_vroom(kind, dict_file, coll)
    assert np.array_equal(enc3, want)  # the same gaps reach the encoder either way


def test_collection_reader_edge_cases(small_corpus):
    """binary_collection::iterator::read (:131-146): empty records are skipped; a last record that claims more values
    than the file holds is cut at the end of the file; a file of nothing but the .docs singleton has no list."""
    kind, dict_file = host.SINGLE_PACKED, small_corpus.dict_file(host.SINGLE_PACKED)
    builder = oracle.OracleBuilder(kind, dict_file)
    lists = [np.array([3, 9, 10, 500], dtype=np.uint32), np.array([7], dtype=np.uint32), np.arange(0, 900, 3, dtype=np.uint32)]
    plain = host.collection_words(lists, num_docs=1000)
    # the same lists with empty records sprinkled in (after record 0, between the lists, at the very end)
    z = np.zeros(1, dtype=np.uint32)
    parts = [np.array([1, 1000], dtype=np.uint32), z, z]
    for v in lists:
        parts += [np.array([v.size], dtype=np.uint32), v, z]
    sprinkled = np.concatenate(parts)
    want, n, ints = builder.encode_collection(plain, docs=True)
    assert (n, ints) == (3, 4 + 1 + 300)
    for words in (plain, sprinkled):
        enc, _u, n2, ints2 = host.encode_collection(kind, dict_file, words, docs=True)
        assert (n2, ints2) == (n, ints) and np.array_equal(enc, want)
        got, n3, ints3 = builder.encode_collection(words, docs=True)
        assert (n3, ints3) == (n, ints) and np.array_equal(got, want)
    # truncated: the last record says 300 values, the file ends after 120 of them
    cut = plain[: plain.size - 180]
    enc, _u, n2, ints2 = host.encode_collection(kind, dict_file, cut, docs=True)
    got, n3, ints3 = builder.encode_collection(cut, docs=True)
    assert (n2, ints2) == (n3, ints3) == (3, 4 + 1 + 120) and np.array_equal(enc, got)
    # nothing behind the singleton
    only = np.array([1, 1000], dtype=np.uint32)
    enc, _u, n2, ints2 = host.encode_collection(kind, dict_file, only, docs=True)
    got, n3, ints3 = builder.encode_collection(only, docs=True)
    assert enc.size == got.size == 0 and (n2, ints2) == (n3, ints3) == (0, 0)


@pytest.mark.parametrize("which,kind", [("single_cases", host.RECTANGULAR), ("single_cases", host.SINGLE_PACKED),
                                        ("multi_cases", host.MULTI_PACKED)])
def test_kat_gap_sequences(which, kind):
    """the hand-assembled KAT dictionaries (tests/golden) and the KATs' expected integers as encoder input: n = 1,
    15/16/17, every run length, 300 zeros, 16- and 32-bit exceptions, all-exception lists ... product == oracle, and the
    bytes decode back (the hand-assembled streams themselves need not be what an optimal parse emits)."""
    dict_file = kat.DICT_FILES[kind]
    builder = oracle.OracleBuilder(kind, dict_file)
    odict = oracle.OracleDict(kind, dict_file)
    for name, _buf, _off, n, expect in kat.cases(which):
        for greedy in ((False, True) if kind != host.MULTI_PACKED else (False,)):
            coll = host.Collection(expect, np.array([n], dtype=np.uint32))
            enc, _ = host.encode_vroom(kind, dict_file, coll, unit_ints=0, greedy=greedy)
            payload = builder.encode_list(expect, greedy=greedy)
            header = oracle.vbyte_encode(n) + oracle.vbyte_encode(int(expect.sum(dtype=np.uint64)) & 0xFFFFFFFF)
            assert enc.tobytes() == header + payload.tobytes(), (name, greedy)
            got, used = odict.decode_list(np.concatenate([payload, np.zeros(8, dtype=np.uint8)]), 0, n)
            assert np.array_equal(got, expect) and used == payload.size, (name, greedy)


def _tiny_single_dict(entries):
    """a single_packed dictionary file holding exactly `entries` (in order), through the ORACLE's builder restatement"""
    return oracle.pack_dictionary(oracle.SINGLE_PACKED, entries)


def test_hash_only_lookup_quirks_single():
    """SURVEY H9, single_dictionary.hpp:154-175 — the map keeps hashes only and `m_map[hash] = i` overwrites:
    (a) a real entry of 16 zeros takes over the hash slot of run codeword 6, so lookup(16 zeros) answers with the entry;
    (b) of two entries with equal integers the LATER one answers. The run codewords themselves are found by counting
    zeros, not through the map (dint_codecs.hpp:215-230), so they stay reachable. Product and oracle agree on the bytes."""
    zeros16 = [0] * 16
    entries = [[5], [5, 6], zeros16, [5, 6], [9, 9, 9, 9]]
    dict_file = _tiny_single_dict(entries)
    builder = oracle.OracleBuilder(oracle.SINGLE_PACKED, dict_file)
    assert builder.lookup([5, 6]) == 7 + 3          # the later duplicate
    assert builder.lookup(zeros16) == 7 + 2         # the real entry, not run codeword 6
    assert builder.lookup([0] * 32) == 5 and builder.lookup([0] * 256) == 2
    assert builder.lookup([1, 2]) == 0xFFFFFFFF
    seqs = [np.array([5, 6] * 9 + [0] * 16 + [9] * 4, dtype=np.uint32), np.array([0] * 40 + [5], dtype=np.uint32),
            np.array(zeros16, dtype=np.uint32), np.array([5, 6, 0, 0, 0], dtype=np.uint32)]
    for g in seqs:
        for greedy in (False, True):
            enc, _ = host.encode_vroom(host.SINGLE_PACKED, dict_file, host.Collection(g, np.array([g.size], dtype=np.uint32)),
                                       unit_ints=0, greedy=greedy)
            hdr = len(oracle.vbyte_encode(g.size)) + len(oracle.vbyte_encode(int(g.sum())))
            assert enc[hdr:].tobytes() == builder.encode_list(g, greedy=greedy).tobytes()
    # greedy takes the longest match first: 16 zeros -> the RUN branch (index 6) before any lookup (dint_codecs.hpp:131-139)
    assert builder.encode_list(np.array(zeros16, dtype=np.uint32), greedy=True).tobytes() == bytes([6, 0])


def test_multi_prepare_stops_reserved_entries_early():
    """multi_dictionary.hpp:201-213: the scan of a dictionary's entries stops `reserved` (7) slots before its end, so
    its last 7 entries are never found by lookup — they become exceptions in the encoder; and only codewords < 256 are
    reachable in 8-bit mode."""
    n = 300
    entries = [[1000 + i] for i in range(n)]
    ctx = [0] * n
    dict_file = oracle.pack_dictionary(oracle.MULTI_PACKED, entries, ctx)
    builder = oracle.OracleBuilder(oracle.MULTI_PACKED, dict_file)
    slots = 7 + n
    for i in range(n):
        code = 7 + i
        found16 = builder.lookup([1000 + i], 0, 16)
        found8 = builder.lookup([1000 + i], 0, 8)
        assert found16 == (code if code < slots - 7 else 0xFFFFFFFF), i
        assert found8 == (code if code < min(256, slots - 7) else 0xFFFFFFFF), i
    assert builder.lookup([1000], 1, 16) == 0xFFFFFFFF  # another dictionary: empty
    g = np.array([1000 + i for i in range(n)][-20:] * 13, dtype=np.uint32)[:256]
    coll = host.Collection(g, np.array([256], dtype=np.uint32))
    enc, _ = host.encode_vroom(host.MULTI_PACKED, dict_file, coll, unit_ints=0)
    hdr = len(oracle.vbyte_encode(256)) + len(oracle.vbyte_encode(int(g.sum(dtype=np.uint64)) & 0xFFFFFFFF))
    assert enc[hdr:].tobytes() == builder.encode_list(g).tobytes()
    got, _ = oracle.OracleDict(oracle.MULTI_PACKED, dict_file).decode_stream(enc, 256)
    assert np.array_equal(got, g)


def test_interpolative_encode_by_hand():
    """interpolative_block::encode (block_codecs.hpp:104-128) on cases small enough to write the bits down.
    n = 1: nothing but (with sum -1) the vbyte of the value. n = 2, values (3, 1), sum given = 4: prefix sums (3, 4);
    write_interpolative(in, 1, 0, 4): val = in[0] = 3, write_int(3, u = 5): b = msb(5) = 2, m = 8 - 5 = 3, 3 >= m ->
    val = 6: write(6 >> 1 = 3, 2 bits) then write(6 & 1 = 0, 1 bit) -> bits 1,1,0 LSB first = 0b011 = 3 -> one byte 0x03."""
    assert oracle.interpolative_encode([7], 7).size == 0
    assert oracle.interpolative_encode([7], 0xFFFFFFFF).tobytes() == bytes([7 | 0x80])
    assert oracle.interpolative_encode([3, 1], 4).tobytes() == bytes([0x03])
    assert oracle.interpolative_encode([3, 1], 0xFFFFFFFF).tobytes() == bytes([4 | 0x80, 0x03])
    # u a power of two: write_int(val, 8): b = 3, m = 16 - 8 = 8, val < m -> 3 plain bits. (5, 2), sum 7 -> val 5 in 3 bits
    assert oracle.interpolative_encode([5, 2], 7).tobytes() == bytes([0x05])
    r = np.random.default_rng(9)
    for n in (1, 2, 3, 17, 100, 255):
        for mag in (1, 7, 20):
            v = r.integers(0, 1 << mag, n, dtype=np.uint64).astype(np.uint32)
            for s in (int(v.sum()), 0xFFFFFFFF):
                enc = oracle.interpolative_encode(v, s)
                got, used = oracle.interpolative_decode(enc, 0, s, n)
                assert np.array_equal(got, v) and used == enc.size


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("greedy", [False, True])
def test_index_lists_bytes_equal_the_oracles(small_corpus, kind, greedy):
    """dict_posting_list::write: vbyte n, block maxima, endpoints, per block docs part + freqs part (full blocks through
    the block coder, short ones interpolative) — product index bytes == oracle, list by list."""
    if greedy and kind == host.MULTI_PACKED:
        pytest.skip("the reference has no greedy multi-dictionary block coder")
    coll = small_corpus.coll
    docids = host.gaps_to_docids(coll)
    freqs = host.synth_freqs(coll.num_postings, 3)
    docs_dict = small_corpus.dict_file(kind)
    freqs_dict = host.build_dictionary(kind, host.Collection(freqs - 1, coll.lens))
    index, offsets = host.build_index(kind, docs_dict, freqs_dict, docids, freqs, coll.lens, greedy=greedy)
    db, fb = oracle.OracleBuilder(kind, docs_dict), oracle.OracleBuilder(kind, freqs_dict)
    b = coll.list_bounds()
    order = np.argsort(coll.lens)
    picks = set(order[-6:].tolist()) | set(order[:: max(1, len(order) // 150)].tolist())
    checked_full = 0
    for i in sorted(picks):
        lo, hi = int(b[i]), int(b[i + 1])
        if hi == lo:
            continue
        want = oracle.posting_list_write(db, fb, docids[lo:hi], freqs[lo:hi], greedy=greedy)
        got = index[int(offsets[i]):int(offsets[i + 1])]
        assert got.size == want.size and np.array_equal(got, want), i
        checked_full += (hi - lo) // 256
    assert checked_full > 100
    # and the collection-file route of the index builder gives the same index
    lists_d = [docids[int(b[i]):int(b[i + 1])] for i in range(len(coll.lens))]
    lists_f = [freqs[int(b[i]):int(b[i + 1])] for i in range(len(coll.lens))]
    idx2, offs2, num_docs = host.build_index_collection(kind, docs_dict, freqs_dict, host.collection_words(lists_d, num_docs=200_000),
                                                        host.collection_words(lists_f), greedy=greedy)
    assert num_docs == 200_000 and np.array_equal(idx2, index) and np.array_equal(offs2, offsets)


def test_block_coder_by_blocks(small_corpus):
    """Coder::encode one block at a time (n = 256 through the dictionary, n < 256 interpolative whatever the coder)."""
    coll = small_corpus.coll
    for kind in KINDS:
        dict_file = small_corpus.dict_file(kind)
        builder, odict = oracle.OracleBuilder(kind, dict_file), oracle.OracleDict(kind, dict_file)
        long_list = int(np.argmax(coll.lens))
        lo = int(coll.list_bounds()[long_list])
        for k in range(0, 12):
            block = coll.gaps[lo + 256 * k: lo + 256 * (k + 1)]
            enc = builder.block_encode(block, int(block.sum()))
            got, used = odict.decode_list(np.concatenate([enc, np.zeros(8, dtype=np.uint8)]), 0, 256)
            assert np.array_equal(got, block) and used == enc.size
        short = coll.gaps[lo: lo + 100]
        assert builder.block_encode(short, int(short.sum())).tobytes() == oracle.interpolative_encode(short, int(short.sum())).tobytes()


@pytest.mark.parametrize("kind", KINDS)
def test_packing_equals_the_oracles(kind):
    """f2's last product-vs-product check: builder::init / append / build / write. The product packs in O(n) with
    hashing (dint/dictionaries.hpp), the oracle restates pack_policy::compact's O(n^2) loops and one std::search per
    entry — same selection in, same dictionary FILE out. The selection: the oracle's own (oracle.Stats.select)."""
    coll = host.synth_collection(120_000, universe=60_000, seed=21)
    multi = kind == host.MULTI_PACKED
    st = oracle.Stats(multi, coll.gaps)
    st.collect_lists(coll.lens)
    entries, ctx, ngrams = [], [], []
    for c in range(st.contexts):
        sel, _passed = st.select(c)
        sel = sel[:3000]  # keep the O(n^2) oracle quick
        for e in sel:
            entries.append(st.ngram(e))
            ctx.append(c)
            ngrams.append((int(e["pos"]), int(e["freq"]), int(e["len"]), c, 0))
    want = oracle.pack_dictionary(kind, entries, ctx if multi else None)
    got = host.pack_dictionary(kind, coll.gaps, np.array(ngrams, dtype=host.NGRAM_DTYPE))
    assert got == want
    # and the packed file is a dictionary both sides read alike: encode with it, product == oracle
    enc, _ = host.encode_vroom(kind, got, coll, unit_ints=0)
    want_enc, _l, _i = oracle.OracleBuilder(kind, want).encode_collection(corpus_words(coll), docs=True)
    assert np.array_equal(enc, want_enc)


def test_packing_prefix_and_duplicate_cases_by_hand():
    """pack_policy::compact on a hand-made selection: duplicates collapse, a proper prefix of a longer entry is dropped
    from the table, the table keeps (size, lexicographic) order behind its 16 zeros, every entry's offset is its FIRST
    occurrence in the table (the zero prefix included)."""
    entries = [[4, 5], [4, 5, 6, 7], [9], [4, 5], [0, 0], [9, 1], [3]]
    f = oracle.pack_dictionary(oracle.SINGLE_PACKED, entries)
    w = np.frombuffer(f, dtype="<u4")
    m_size, n_off, n_tab = int(w[0]), int(w[1]), int(w[2])
    offsets, table = w[3:3 + n_off], w[3 + n_off:3 + n_off + n_tab]
    assert m_size == 7 + len(entries) and n_off == 7 + len(entries)
    # survivors sorted by (size, lex): [3] | [0,0] [9,1] | [4,5,6,7]   ([9] and [4,5] are prefixes; duplicates gone)
    assert table.tolist() == [0] * 16 + [3] + [0, 0] + [9, 1] + [4, 5, 6, 7]
    size_off = [(int(o >> 24) + 1, int(o & 0xFFFFFF)) for o in offsets[7:]]
    assert size_off == [(2, 21), (4, 21), (1, 19), (2, 21), (2, 0), (2, 19), (1, 16)]
    got = host.pack_dictionary(host.SINGLE_PACKED, np.concatenate([np.array(e, dtype=np.uint32) for e in entries]),
                               np.array([(p, 1, l, 0, 0) for p, l in zip(np.cumsum([0] + [len(e) for e in entries[:-1]]),
                                                                          [len(e) for e in entries])], dtype=host.NGRAM_DTYPE))
    assert got == f


def _run(tool, *args, cwd):
    r = subprocess.run([os.path.join(BIN, tool), *args], cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout.strip().splitlines()


@pytest.mark.parametrize("kind", KINDS)
def test_ingest_tools_on_a_written_collection(tmp_path, kind):
    """dint_build_dict + dint_encode + dint_create_freq_index on a written .docs / .freqs pair, with the reference's
    argument shapes (vroom_env/encode.cpp:283-310, src/create_freq_index.cpp:112-131) and dictionary file naming
    (dict_freq_index.hpp:141-147): the encoded files equal the oracle's `encode` over the same words with the tools'
    dictionaries; the index file's lists equal oracle.posting_list_write."""
    for tool in ("dint_build_dict", "dint_encode", "dint_create_freq_index"):
        if not os.path.exists(os.path.join(BIN, tool)):
            subprocess.run(["make", "-C", os.path.join(ROOT, "dint_amd", "csrc"), "host-tools"], check=True)
    coll = host.synth_collection(150_000, universe=80_000, seed=31)
    docids = host.gaps_to_docids(coll)
    freqs = host.synth_freqs(coll.num_postings, 5)
    b = coll.list_bounds()
    lists_d = [docids[int(b[i]):int(b[i + 1])] for i in range(len(coll.lens))]
    lists_f = [freqs[int(b[i]):int(b[i + 1])] for i in range(len(coll.lens))]
    base = str(tmp_path / "coll")
    host.write_collection(base, lists_d, lists_f, num_docs=80_000)
    type_name = TYPE_OF[kind]
    suffix = {host.RECTANGULAR: "rectangular", host.SINGLE_PACKED: "single_packed", host.MULTI_PACKED: "multi_packed"}[kind]

    out = _run("dint_build_dict", type_name, base, "--threads", "4", cwd=tmp_path)
    lines = [json.loads(x) for x in out]
    assert [x["built"] for x in lines] == ["true", "true"]
    dict_docs = tmp_path / f"dict.coll.docs.{suffix}.DSF-65536-16"
    dict_freqs = tmp_path / f"dict.coll.freqs.{suffix}.DSF-65536-16"
    assert dict_docs.exists() and dict_freqs.exists()
    # "build or load": a second run leaves the files alone
    again = [json.loads(x) for x in _run("dint_build_dict", type_name, base, cwd=tmp_path)]
    assert [x["built"] for x in again] == ["false", "false"]
    # the dictionary is the library's construction over the same lists
    assert dict_docs.read_bytes() == host.build_dictionary(kind, coll)

    for ext, dict_path, is_docs in ((".docs", dict_docs, True), (".freqs", dict_freqs, False)):
        enc_path = tmp_path / f"enc{ext}.bin"
        units_path = tmp_path / f"units{ext}.bin"
        line = json.loads(_run("dint_encode", type_name, base + ext, "--dict", str(dict_path), "--out", str(enc_path),
                               "--units", str(units_path), "--unit-ints", "2048", "--threads", "4", cwd=tmp_path)[-1])
        assert set(line) == {"filename", "num_sequences", "num_integers", "type", "GiB", "bpi"}  # encode.cpp:49-58
        words = np.fromfile(base + ext, dtype=np.uint32)
        want, lists, ints = oracle.OracleBuilder(kind, dict_path.read_bytes()).encode_collection(words, docs=is_docs)
        assert int(line["num_sequences"]) == lists and int(line["num_integers"]) == ints == coll.num_postings
        got = np.fromfile(enc_path, dtype=np.uint8)
        assert np.array_equal(got, want)
        units = np.fromfile(units_path, dtype=host.UNIT_DTYPE)
        assert int(units["n"].sum()) == ints and int(units["n"].max()) <= 2048 + 256
        dec, _ = oracle.OracleDict(kind, dict_path.read_bytes()).decode_stream(got, ints)
        assert np.array_equal(dec, coll.gaps if is_docs else freqs - 1)

    idx_path = tmp_path / "coll.index"
    line = json.loads(_run("dint_create_freq_index", type_name, base, str(idx_path), "--threads", "4", cwd=tmp_path)[-1])
    assert line["type"] == type_name and line["sequences"] == len(coll.lens) and line["postings"] == coll.num_postings
    f = host.read_index_file(str(idx_path))
    assert f["kind"] == kind and f["num_docs"] == 80_000 and f["docs_dict"] == dict_docs.read_bytes()
    db, fb = oracle.OracleBuilder(kind, f["docs_dict"]), oracle.OracleBuilder(kind, f["freqs_dict"])
    od, of = oracle.OracleDict(kind, f["docs_dict"]), oracle.OracleDict(kind, f["freqs_dict"])
    for i in list(np.argsort(coll.lens)[-3:]) + list(range(0, len(coll.lens), max(1, len(coll.lens) // 40))):
        lo, hi = int(b[i]), int(b[i + 1])
        lst = f["index"][int(f["offsets"][i]):int(f["offsets"][i + 1])]
        assert np.array_equal(lst, oracle.posting_list_write(db, fb, docids[lo:hi], freqs[lo:hi]))
        d, fr = oracle.posting_list_decode(od, of, f["index"], int(f["offsets"][i]))
        assert np.array_equal(d, docids[lo:hi]) and np.array_equal(fr, freqs[lo:hi])


def test_tools_argument_errors(tmp_path):
    """unknown type: logged, exit 0 (encode.cpp:324-328); a missing --dict or a file that is neither .docs nor .freqs:
    an error (encode.cpp:138-140, :165-167)."""
    (tmp_path / "x.docs").write_bytes(np.array([1, 10, 2, 3, 4], dtype=np.uint32).tobytes())
    r = subprocess.run([os.path.join(BIN, "dint_encode"), "no_such_type", str(tmp_path / "x.docs")], capture_output=True, text=True)
    assert r.returncode == 0 and "unknown type" in r.stderr
    r = subprocess.run([os.path.join(BIN, "dint_encode"), "single_packed_dint", str(tmp_path / "x.docs")], capture_output=True, text=True)
    assert r.returncode == 1 and "dictionary_filename must be specified" in r.stderr
    (tmp_path / "x.bin").write_bytes(b"\0" * 8)
    r = subprocess.run([os.path.join(BIN, "dint_encode"), "single_packed_dint", str(tmp_path / "x.bin"), "--dict", str(tmp_path / "x.docs")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "unsupported file format" in r.stderr


def test_committed_encoder_vectors():
    """tests/golden/encoder_vectors.json (make_encoder_vectors.py): the bytes the oracle's encoder emitted for the KAT
    dictionaries' gap sequences and a few small posting lists when the file was made — a regression pin (not reference
    outputs): the oracle still emits them, and so does the product."""
    import hashlib

    with open(os.path.join(ROOT, "tests", "golden", "encoder_vectors.json")) as f:
        vec = json.load(f)
    builders = {k: oracle.OracleBuilder(k, kat.DICT_FILES[k]) for k in KINDS}
    gaps_of = {(w, name): expect for w in ("single_cases", "multi_cases") for name, _b, _o, _n, expect in kat.cases(w)}
    assert len(vec["lists"]) > 50 and len(vec["posting_lists"]) == 8

    def same(payload: bytes, v) -> bool:
        return len(payload) == v["bytes"] and (payload.hex() == v["payload"] if "payload" in v else
                                               hashlib.sha256(payload).hexdigest() == v["payload_sha256"])

    for v in vec["lists"]:
        gaps = gaps_of[(v["which"], v["case"])]
        assert same(builders[v["kind"]].encode_list(gaps, greedy=v["greedy"]).tobytes(), v), v["case"]
        enc, _ = host.encode_vroom(v["kind"], kat.DICT_FILES[v["kind"]], host.Collection(gaps, np.array([gaps.size], dtype=np.uint32)),
                                   unit_ints=0, greedy=v["greedy"])
        hdr = len(oracle.vbyte_encode(gaps.size)) + len(oracle.vbyte_encode(int(gaps.sum(dtype=np.uint64)) & 0xFFFFFFFF))
        assert same(enc[hdr:].tobytes(), v), v["case"]
    for v in vec["posting_lists"]:
        d, fr = np.array(v["docids"], dtype=np.uint32), np.array(v["freqs"], dtype=np.uint32)
        b = builders[v["kind"]]
        assert oracle.posting_list_write(b, b, d, fr).tobytes().hex() == v["bytes"]
        idx, offs = host.build_index(v["kind"], kat.DICT_FILES[v["kind"]], kat.DICT_FILES[v["kind"]], d, fr, np.array([d.size], dtype=np.uint32))
        assert idx.tobytes().hex() == v["bytes"]

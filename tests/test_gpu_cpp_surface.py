"""The C++ operator surface (dint/coders.hpp) and the dint_decode CLI on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from dint_amd import host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = {host.RECTANGULAR: "single_rect_dint", host.SINGLE_PACKED: "single_packed_dint",
         host.MULTI_PACKED: "multi_packed_dint"}


@pytest.fixture(scope="module")
def roundtrip_binary(tmp_path_factory):
    exe = tmp_path_factory.mktemp("cpp") / "coder_roundtrip"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/dint_amd/csrc/host",
                    os.path.join(ROOT, "tests", "cpp", "coder_roundtrip.cpp"), "-o", str(exe),
                    f"-L{ROOT}/dint_amd", "-ldint_hip", f"-Wl,-rpath,{ROOT}/dint_amd"], check=True)
    return str(exe)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_coder_call_shape_round_trip(roundtrip_binary, small_corpus, tmp_path, mode):
    kind = {0: host.RECTANGULAR, 1: host.SINGLE_PACKED, 2: host.MULTI_PACKED, 3: host.SINGLE_PACKED}[mode]
    (tmp_path / "dict.bin").write_bytes(small_corpus.dict_file(kind))
    bounds = small_corpus.coll.list_bounds()
    longest = int(np.argmax(small_corpus.coll.lens))
    small_corpus.coll.gaps[int(bounds[longest]):int(bounds[longest + 1])].tofile(tmp_path / "gaps.bin")
    r = subprocess.run([roundtrip_binary, str(mode), str(tmp_path / "dict.bin"), str(tmp_path / "gaps.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


@pytest.mark.parametrize("kind", list(TYPES))
def test_decode_cli_reports_the_reference_keys(small_corpus, tmp_path, kind):
    enc, _ = small_corpus.encoded(kind)
    (tmp_path / "test.bin").write_bytes(enc.tobytes())
    (tmp_path / "dict.bin").write_bytes(small_corpus.dict_file(kind))
    exe = os.path.join(ROOT, "dint_amd", "bin", "dint_decode")
    r = subprocess.run([exe, TYPES[kind], str(tmp_path / "test.bin"), "--dict", str(tmp_path / "dict.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("filename", "num_sequences", "num_integers", "type", "tot_elapsed_time", "ns_x_int", "ints_x_sec"):
        assert key in line                                  # vroom_env/statistics.hpp:26-34
    assert int(line["num_integers"]) == small_corpus.coll.num_postings
    assert int(line["num_sequences"]) == np.count_nonzero(small_corpus.coll.lens)
    assert line["type"] == TYPES[kind] and int(line["ints_x_sec"]) > 0


def test_decode_cli_on_the_readme_shaped_collection(tmp_path):
    """BASELINE config 1: `decode single_rect_dint` on the reference's test collection — whose .docs file is not in the
    checkout: a seeded stand-in of the shape README.md:53 states (10 000 documents, 113 306 lists,
    3 327 520 postings), every list encoded, the tool's output checked against the encoder's input."""
    coll = host.readme_test_collection(seed=1)
    assert (len(coll.lens), coll.num_postings) == (113_306, 3_327_520)
    dict_file = host.build_dictionary(host.RECTANGULAR, coll)
    enc, _ = host.encode_vroom(host.RECTANGULAR, dict_file, coll, unit_ints=8192)
    (tmp_path / "test_collection.bin").write_bytes(enc.tobytes())
    (tmp_path / "dict.bin").write_bytes(dict_file)
    (tmp_path / "gaps.bin").write_bytes(coll.gaps.tobytes())
    exe = os.path.join(ROOT, "dint_amd", "bin", "dint_decode")
    r = subprocess.run([exe, "single_rect_dint", str(tmp_path / "test_collection.bin"), "--dict", str(tmp_path / "dict.bin"),
                        "--check", str(tmp_path / "gaps.bin")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert int(line["num_sequences"]) == 113_306 and int(line["num_integers"]) == 3_327_520
    assert line["type"] == "single_rect_dint" and line["bit_exact"] == "true"
    # and the tool notices when the integers are not the expected ones
    wrong = coll.gaps.copy()
    wrong[123_456] ^= 1
    (tmp_path / "gaps.bin").write_bytes(wrong.tobytes())
    r = subprocess.run([exe, "single_rect_dint", str(tmp_path / "test_collection.bin"), "--dict", str(tmp_path / "dict.bin"),
                        "--check", str(tmp_path / "gaps.bin")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and json.loads(r.stdout.strip().splitlines()[-1])["bit_exact"] == "false"


@pytest.fixture(scope="module")
def walker_binary(tmp_path_factory):
    exe = tmp_path_factory.mktemp("cpp") / "block_walker"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/dint_amd/csrc/host",
                    os.path.join(ROOT, "tests", "cpp", "block_walker.cpp"), "-o", str(exe),
                    f"-L{ROOT}/dint_amd", "-ldint_hip", f"-Wl,-rpath,{ROOT}/dint_amd"], check=True)
    return str(exe)


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_block_coder_under_a_posting_list_walker(walker_binary, small_corpus, tmp_path, kind):
    """The device block Coders instantiate under a dict_posting_list-shaped enumerator (tests/cpp/block_walker.cpp):
    full blocks and an interpolative tail, docs with their sum_of_values hint, freqs with -1; every posting equals
    the oracle's document_enumerator walk and the builder's input."""
    import oracle
    from test_index_cpu import get_index

    ix = get_index(small_corpus, kind)
    od, of = oracle.OracleDict(kind, ix.docs_dict), oracle.OracleDict(kind, ix.freqs_dict)
    # lists with several full blocks and a tail, a pure tail (< 256), and an exact multiple of 256 if there is one
    lens = np.asarray(ix.lens)
    picks = [int(np.flatnonzero((lens > 600) & (lens % 256 != 0))[0]), int(np.flatnonzero((lens > 3) & (lens < 256))[0])]
    exact = np.flatnonzero((lens >= 256) & (lens % 256 == 0))
    if len(exact):
        picks.append(int(exact[0]))
    (tmp_path / "docs.dict").write_bytes(ix.docs_dict)
    (tmp_path / "freqs.dict").write_bytes(ix.freqs_dict)
    for i in picks:
        lo, hi = int(ix.offsets[i]), int(ix.offsets[i + 1])
        d, f = oracle.posting_list_decode(od, of, ix.bytes, lo)
        a, b = int(ix.bounds[i]), int(ix.bounds[i + 1])
        assert np.array_equal(d, ix.docids[a:b]) and np.array_equal(f, ix.freqs[a:b])
        ix.bytes[lo:hi].tofile(tmp_path / "list.bin")
        d.astype(np.uint32).tofile(tmp_path / "docids.bin")
        f.astype(np.uint32).tofile(tmp_path / "freqs.bin")
        r = subprocess.run([walker_binary, str(kind), str(tmp_path / "docs.dict"), str(tmp_path / "freqs.dict"),
                            str(tmp_path / "list.bin"), str(tmp_path / "docids.bin"), str(tmp_path / "freqs.bin")],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip() == "ok", f"list {i} (n={lens[i]}): " + r.stdout + r.stderr
        # scopes destroyed out of order (a container's order), a docs-only scope falling back for the freqs parts
        r = subprocess.run([walker_binary, str(kind), str(tmp_path / "docs.dict"), str(tmp_path / "freqs.dict"),
                            str(tmp_path / "list.bin"), str(tmp_path / "docids.bin"), str(tmp_path / "freqs.bin"), "scopes"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip() == "ok", f"list {i} (n={lens[i]}) scopes: " + r.stdout + r.stderr


def test_decode_block_host_call_shape(small_corpus):
    """dint_decode_block_host directly: consumed bytes are the block's bytes, for docs (sum hint) and freqs (-1) parts."""
    from dint_amd import device
    from test_index_cpu import get_index

    kind = host.SINGLE_PACKED
    ix = get_index(small_corpus, kind)
    blocks, _ = device.index_posting_lists(ix.bytes, ix.offsets)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
    seen = set()
    for b in range(len(blocks)):
        n = int(blocks["n"][b])
        key = (n == 256, n < 16)
        if key in seen and len(seen) >= 3:
            continue
        seen.add(key)
        at = int(blocks["in_off"][b])
        gaps_sum = (int(blocks["max"][b]) - int(blocks["base"][b]) - (n - 1)) & 0xFFFFFFFF
        docs, used = device.decode_block(dd, padded, at, gaps_sum, n)
        lo = int(blocks["out_off"][b])
        want = ix.docids[lo:lo + n].astype(np.int64)
        gaps = np.diff(np.r_[int(blocks["base"][b]) - 1, want]) - 1
        assert np.array_equal(docs, gaps.astype(np.uint32))
        freqs, used_f = device.decode_block(fd, padded, at + used, 0xFFFFFFFF, n)
        assert np.array_equal(freqs + 1, ix.freqs[lo:lo + n])
        nxt = int(blocks["in_off"][b + 1]) if b + 1 < len(blocks) and blocks["list"][b + 1] == blocks["list"][b] else None
        if nxt is not None:
            assert at + used + used_f == nxt


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_block_coder_under_a_list_scope(walker_binary, tmp_path, kind):
    """A usable per-block drop-in: inside a Coder::list_scope the list is decoded ONCE on the device and every
    Coder::decode of the walker — the reference's end-less call shape — is a memcpy. A 10^5-posting list (390 full blocks
    and an interpolative tail): every posting right, the walk takes milliseconds, and the library makes no device or
    pinned allocation during it (the dictionaries' workspaces are warm). The same walk with one launch sequence per block
    (mode noend: bounded by Coder::readable_end()) is checked for equality too."""
    import oracle

    r = np.random.default_rng(5)
    n = 100_123
    docids = np.cumsum(r.geometric(0.02, n).astype(np.uint64)).astype(np.uint32)
    freqs = r.geometric(0.6, n).astype(np.uint32)
    gaps = host.docids_to_gaps(docids)
    coll = host.Collection(gaps, np.array([n], dtype=np.uint32))
    fcoll = host.Collection(freqs - 1, np.array([n], dtype=np.uint32))
    docs_dict, freqs_dict = host.build_dictionary(kind, coll), host.build_dictionary(kind, fcoll)
    index, offsets = host.build_index(kind, docs_dict, freqs_dict, docids, freqs, np.array([n], dtype=np.uint32))
    d, f = oracle.posting_list_decode(oracle.OracleDict(kind, docs_dict), oracle.OracleDict(kind, freqs_dict), index, 0)
    assert np.array_equal(d, docids) and np.array_equal(f, freqs)
    (tmp_path / "docs.dict").write_bytes(docs_dict)
    (tmp_path / "freqs.dict").write_bytes(freqs_dict)
    index[int(offsets[0]):int(offsets[1])].tofile(tmp_path / "list.bin")
    docids.tofile(tmp_path / "docids.bin")
    freqs.tofile(tmp_path / "freqs.bin")
    args = [walker_binary, str(kind), str(tmp_path / "docs.dict"), str(tmp_path / "freqs.dict"), str(tmp_path / "list.bin"),
            str(tmp_path / "docids.bin"), str(tmp_path / "freqs.bin")]
    best = None
    for _ in range(3):
        res = subprocess.run(args + ["scope"], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0 and res.stdout.splitlines()[0] == "ok", res.stdout + res.stderr
        fields = res.stdout.splitlines()[1].split()
        ms, allocations = float(fields[1]), int(fields[3])
        assert allocations == 0 and int(fields[5]) == n
        best = ms if best is None else min(best, ms)
    print(f"walk of {n} postings inside a list_scope: {best:.2f} ms")
    assert best < 5.0   # (about 1 ms on an idle box; one launch sequence per block takes 40 times that)
    res = subprocess.run(args + ["noend"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip() == "ok", res.stdout + res.stderr


def test_list_cache_serves_the_block_coder_call(small_corpus):
    """dint_list_cache_*: for every block of a list, the docs part and the freqs part equal dint_decode_block_host's and
    the consumed bytes chain through the list; an offset where no part starts is an error."""
    from dint_amd import device
    from test_index_cpu import get_index

    kind = host.SINGLE_PACKED
    ix = get_index(small_corpus, kind)
    blocks, _ = device.index_posting_lists(ix.bytes, ix.offsets)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    lens = np.asarray(ix.lens)
    for i in (int(np.argmax(lens)), int(np.flatnonzero((lens > 3) & (lens < 256))[0])):
        lo, hi = int(ix.offsets[i]), int(ix.offsets[i + 1])
        cache = device.ListCache(dd, fd, ix.bytes[lo:hi])
        mine = blocks[blocks["list"] == i]
        padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
        before = None
        for b in mine[:: max(1, len(mine) // 12)]:
            n, at = int(b["n"]), int(b["in_off"])
            docs, used = cache.decode(at - lo, n)
            gaps_sum = (int(b["max"]) - int(b["base"]) - (n - 1)) & 0xFFFFFFFF
            want, want_used = device.decode_block(dd, padded, at, gaps_sum, n)
            assert np.array_equal(docs, want) and used == want_used
            fr, used_f = cache.decode(at - lo + used, n)
            want_f, want_used_f = device.decode_block(fd, padded, at + used, 0xFFFFFFFF, n)
            assert np.array_equal(fr, want_f) and used_f == want_used_f
            a = int(b["out_off"])
            assert np.array_equal(fr + 1, ix.freqs[a:a + n])
            if before is None:
                before = device.alloc_count()   # (both dictionaries' host-call workspaces are warm from here on)
        assert device.alloc_count() == before   # no allocation per call: one block at a time or from the cache
        with pytest.raises(device.DintError):
            cache.decode(1, 3)
        cache.close()


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_a_written_collection_through_every_tool(tmp_path, kind):
    """The chain a user of the reference runs, tool by tool, on a written .docs / .freqs pair (BASELINE config 5 end to end):
    dint_build_dict -> dint_encode -> dint_decode --check (the gaps) ; dint_create_freq_index -> dint_queries and / and_freq
    on a query log from stdin: the totals are the oracle's and_query over the same index, the stats line carries the
    reference's keys (src/queries.cpp:45-59)."""
    import oracle
    from queries import reference_queries

    coll = host.synth_collection(400_000, universe=150_000, seed=41)
    docids = host.gaps_to_docids(coll)
    freqs = host.synth_freqs(coll.num_postings, 9)
    b = coll.list_bounds()
    base = str(tmp_path / "c")
    host.write_collection(base, [docids[int(b[i]):int(b[i + 1])] for i in range(len(coll.lens))],
                          [freqs[int(b[i]):int(b[i + 1])] for i in range(len(coll.lens))], num_docs=int(docids.max()) + 1)
    t = TYPES[kind]
    suffix = {host.SINGLE_PACKED: "single_packed", host.MULTI_PACKED: "multi_packed"}[kind]
    bin_ = lambda name: os.path.join(ROOT, "dint_amd", "bin", name)
    run = lambda *a, **kw: subprocess.run(list(a), cwd=tmp_path, capture_output=True, text=True, timeout=900, **kw)
    r = run(bin_("dint_build_dict"), t, base, "--threads", "4")
    assert r.returncode == 0, r.stderr
    dict_docs = str(tmp_path / f"dict.c.docs.{suffix}.DSF-65536-16")
    r = run(bin_("dint_encode"), t, base + ".docs", "--dict", dict_docs, "--out", str(tmp_path / "docs.enc"), "--threads", "4")
    assert r.returncode == 0, r.stderr
    coll.gaps.tofile(tmp_path / "gaps.bin")
    r = run(bin_("dint_decode"), t, str(tmp_path / "docs.enc"), "--dict", dict_docs, "--check", str(tmp_path / "gaps.bin"))
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["bit_exact"] == "true", r.stderr
    r = run(bin_("dint_create_freq_index"), t, base, str(tmp_path / "c.index"), "--threads", "4")
    assert r.returncode == 0, r.stderr
    qs = [q for q in reference_queries(len(coll.lens))[:120]]
    log = "\n".join(" ".join(str(int(x)) for x in q) for q in qs) + "\n"
    r = run(bin_("dint_queries"), t, "and:and_freq:wand", str(tmp_path / "c.index"), "--runs", "3", "--batch", input=log)
    assert r.returncode == 0, r.stderr
    assert "Unsupported query type: wand" in r.stderr  # src/queries.cpp:108-110
    f = host.read_index_file(str(tmp_path / "c.index"))
    oi = oracle.OracleIndex(oracle.OracleDict(kind, f["docs_dict"]), f["index"], f["offsets"], f["num_docs"])
    want = sum(oi.and_query(q) for q in qs)
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 4 and int(lines[0]) == 3 * want and int(lines[2]) == 3 * want
    for text, name in ((lines[1], "and"), (lines[3], "and_freq")):
        line = json.loads(text)
        assert line["type"] == t and line["query"] == name and line["avg"] > 0 and line["q50"] <= line["q95"]
        assert set(("type", "query", "avg", "q50", "q90", "q95")) <= set(line) and line["batch_us_per_query"] > 0

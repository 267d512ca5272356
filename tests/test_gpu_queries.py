"""AND queries on the GPU through the C ABI (dint_and_queries): result counts identical to the
oracle's and_query and to plain set intersection of the index builder's input."""
import numpy as np
import pytest

import oracle
from dint_amd import host
from queries import ReadmeIndex, heavy_queries, intersect, intersect_freqs, reference_queries
from test_index_cpu import get_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    import torch

    assert torch.cuda.is_available()
    from dint_amd import device as dev

    return dev


@pytest.fixture(autouse=True)
def _options_back_to_default(device):
    yield
    device.reset_options()


def _query_index(device, ix, kind):
    dd = device.Dictionary(kind, ix.docs_dict)
    return device.QueryIndex(dd, ix.bytes, ix.offsets)


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.RECTANGULAR, host.MULTI_PACKED])
@pytest.mark.parametrize("corpus_name", ["small_corpus", "dense_corpus", "sparse_corpus"])
def test_batch_matches_oracle(device, request, kind, corpus_name):
    corpus = request.getfixturevalue(corpus_name)
    ix = get_index(corpus, kind)
    qi = _query_index(device, ix, kind)
    qs = reference_queries(len(ix.lens)) + heavy_queries(ix.lens, 200)
    got = qi.and_queries(qs)
    want = np.array([intersect(ix.docids, ix.bounds, q) for q in qs], dtype=np.uint64)
    assert np.array_equal(got, want)
    assert int(want.sum()) > 1000
    od = oracle.OracleDict(kind, ix.docs_dict)
    nd = int(ix.docids.max()) + 1
    for i in range(0, len(qs), 7):
        assert int(got[i]) == oracle.and_query(od, ix.bytes, ix.offsets, nd, qs[i])
    qi.close()


@pytest.mark.parametrize("batch_fused", [1, 0])
@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
@pytest.mark.parametrize("corpus_name", ["small_corpus", "sparse_corpus"])
def test_a_call_of_small_queries(device, request, kind, corpus_name, batch_fused):
    """A call whose queries all have at most 16 candidate pages is ONE launch, a workgroup per query walking the query's whole
    chain (query_batch_body: per-workgroup claim sets, per-query stretches of the page buffers); query_batch_fused = 0 is the
    round-per-launch batch form. The reference's log restricted to such queries — more queries than workgroups, so a
    workgroup walks several and finds the claims the one before released — then a mixed call (the other form), a single
    query, and the small call again: the forms leave nothing behind for each other."""
    device.set_option("query_batch_fused", batch_fused)
    corpus = request.getfixturevalue(corpus_name)
    ix = get_index(corpus, kind)
    qi = _query_index(device, ix, kind)
    every = reference_queries(len(ix.lens)) + heavy_queries(ix.lens, 60, seed=3)
    small = [q for q in every if len(set(q)) >= 2 and min(int(ix.lens[t]) for t in q) <= 16 * 256]
    assert len(small) >= 250, "this corpus has too few lists of at most 16 blocks"
    small = (small * 3)[:900] + [[small[0][0]] * 3, [], small[1][:1]]   # (+ a repeated term, an empty query, a single term)
    assert len(small) > 600
    want = np.array([intersect(ix.docids, ix.bounds, q) for q in small], dtype=np.uint64)
    assert np.array_equal(qi.and_queries(small), want) and int(want.sum()) > 100
    mixed = every[:200]
    assert np.array_equal(qi.and_queries(mixed), np.array([intersect(ix.docids, ix.bounds, q) for q in mixed], dtype=np.uint64))
    assert int(qi.and_queries([small[5]])[0]) == int(want[5])
    assert np.array_equal(qi.and_queries(small), want)
    qi.close()


def test_one_query_at_a_time_and_workspace_reuse(device, small_corpus):
    """op_perftest runs the queries one by one (src/queries.cpp:15-61)."""
    kind = host.SINGLE_PACKED
    ix = get_index(small_corpus, kind)
    qi = _query_index(device, ix, kind)
    qs = heavy_queries(ix.lens, 40, seed=9)
    batch = qi.and_queries(qs)
    for q, want in zip(qs, batch):
        assert int(qi.and_queries([q])[0]) == int(want)
    assert np.array_equal(qi.and_queries(qs), batch)


def test_edges(device, dense_corpus):
    kind = host.SINGLE_PACKED
    ix = get_index(dense_corpus, kind)
    qi = _query_index(device, ix, kind)
    longest, shortest = int(np.argmax(ix.lens)), int(np.argmin(ix.lens))
    got = qi.and_queries([[], [longest], [longest] * 4, [shortest, longest], [shortest]])
    assert got[0] == 0
    assert got[1] == ix.lens[longest] and got[2] == ix.lens[longest]
    assert got[3] == intersect(ix.docids, ix.bounds, [shortest, longest])
    assert got[4] == ix.lens[shortest]
    assert qi.and_queries([]).size == 0
    assert np.array_equal(qi.and_queries([[], []]), [0, 0])
    with pytest.raises(device.DintError):
        qi.and_queries([[len(ix.lens)]])  # no such list


def test_disjoint_and_identical_lists(device):
    """Hand-made lists: disjoint ranges (every probe misses past the last block), interleaved
    evens/odds (every probe lands in a block and misses), and a list ANDed with a superset."""
    kind = host.SINGLE_PACKED
    a = np.arange(0, 3000, dtype=np.uint32)                  # 0..2999
    b = np.arange(5000, 9000, dtype=np.uint32)               # disjoint, beyond a
    ev = np.arange(0, 20000, 2, dtype=np.uint32)
    od_ = np.arange(1, 20000, 2, dtype=np.uint32)
    sup = np.arange(0, 20000, dtype=np.uint32)
    lists = [a, b, ev, od_, sup]
    docids = np.concatenate(lists)
    lens = np.array([len(x) for x in lists], dtype=np.uint32)
    gaps = np.concatenate([host.docids_to_gaps(x) for x in lists])
    coll = host.Collection(gaps, lens)
    freqs = np.ones(docids.size, dtype=np.uint32)
    dd = host.build_dictionary(kind, coll)
    fd = host.build_dictionary(kind, host.Collection(freqs - 1, lens))
    idx, offs = host.build_index(kind, dd, fd, docids, freqs, lens)
    qi = device.QueryIndex(device.Dictionary(kind, dd), idx, offs)
    got = qi.and_queries([[0, 1], [1, 0], [2, 3], [2, 4], [3, 4, 2], [0, 4], [0, 2], [1, 3, 4]])
    assert list(got) == [0, 0, 0, len(ev), 0, len(a), 1500, 2000]


@pytest.mark.parametrize("lean_pages", ["0", "1000000000"])
@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_both_page_decode_forms(device, small_corpus, kind, lean_pages):
    """A round's pages are decoded by one launch (few pages: decode_*_query_kernel) or by three (prepare, the
    scheduled decode kernel, fix-up); the option query_lean_pages moves the switch — both forms, forced, on batches
    and on single queries."""
    device.set_option("query_lean_pages", int(lean_pages))
    ix = get_index(small_corpus, kind)
    qi = _query_index(device, ix, kind)
    qs = reference_queries(len(ix.lens))[:300] + heavy_queries(ix.lens, 100, seed=5)
    want = np.array([intersect(ix.docids, ix.bounds, q) for q in qs], dtype=np.uint64)
    assert np.array_equal(qi.and_queries(qs), want)
    for q, w in list(zip(qs, want))[::9]:
        assert int(qi.and_queries([q])[0]) == int(w)
    qi.close()


@pytest.mark.parametrize("fused_pages", ["0", "1", "4", "4, inputs copied on the stream"])
@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_whole_query_in_one_launch(device, small_corpus, kind, fused_pages):
    """A query of a few candidate pages runs as ONE launch of one workgroup that walks the whole chain — candidates, every
    round's pages, every round's tail (query_fused_body); the option query_fused_pages moves the switch (0: never — the
    round-per-launch form). Single queries and the freqs variant (whose counting half takes the same path), all forms
    equal to the plain intersection. The one workgroup fetches the call's inputs from pinned host memory itself
    (query_fused_copy = 0: a copy on the stream in front of the launch, as the other forms have it)."""
    if "," in fused_pages:
        device.set_option("query_fused_copy", 0)
        fused_pages = fused_pages.split(",")[0]
    device.set_option("query_fused_pages", int(fused_pages))
    device.set_option("query_tail_pages", 4)
    ix = get_index(small_corpus, kind)
    qi = _query_index(device, ix, kind)
    fd = device.Dictionary(kind, ix.freqs_dict)
    qs = reference_queries(len(ix.lens))[:120] + heavy_queries(ix.lens, 40, seed=9)
    for q in qs:
        want = intersect(ix.docids, ix.bounds, q)
        assert int(qi.and_queries([q])[0]) == want, q
    for q in qs[::7]:
        counts, sums, _ = qi.and_queries_with_freqs(fd, [q])
        assert int(counts[0]) == intersect(ix.docids, ix.bounds, q)
    assert np.array_equal(qi.and_queries(qs), np.array([intersect(ix.docids, ix.bounds, q) for q in qs], dtype=np.uint64))
    qi.close()


@pytest.mark.parametrize("tail_pages", ["0", "4", "64"])
@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_round_tail_over_several_workgroups(device, kind, tail_pages):
    """A small call runs a whole round per launch (round_tail): the workgroup of the page decode that finishes last
    probes and searches. A short rarest list whose postings fall into every block of two long lists makes the
    rounds' page decodes launches of several workgroups (the fenced hand-over), a rarest list of one short block
    the launch of one; query_tail_pages = 0 is the same call without the tail."""
    device.set_option("query_tail_pages", int(tail_pages))
    r = np.random.default_rng(5)
    long_a = np.arange(0, 60000, 3, dtype=np.uint32)                      # 20000 postings, 79 blocks
    long_b = np.unique(r.integers(0, 60000, 30000)).astype(np.uint32)     # ~ 24000 postings
    rare = np.arange(0, 60000, 75, dtype=np.uint32)                       # 800 postings: 4 pages, spread over every block
    tiny = np.arange(30, 60000, 1500, dtype=np.uint32)                    # 40 postings: one short block
    lists = [long_a, long_b, rare, tiny]
    docids = np.concatenate(lists)
    lens = np.array([len(x) for x in lists], dtype=np.uint32)
    bounds = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    gaps = np.concatenate([host.docids_to_gaps(x) for x in lists])
    coll = host.Collection(gaps, lens)
    freqs = np.ones(docids.size, dtype=np.uint32)
    dd = host.build_dictionary(kind, coll)
    fd = host.build_dictionary(kind, host.Collection(freqs - 1, lens))
    idx, offs = host.build_index(kind, dd, fd, docids, freqs, lens)
    qi = device.QueryIndex(device.Dictionary(kind, dd), idx, offs)
    qs = [[2, 0], [2, 1], [2, 0, 1], [3, 0, 1], [3, 2], [3, 2, 0, 1], [0, 1]]
    want = [intersect(docids, bounds, q) for q in qs]
    assert want[0] == len(rare) and want[2] > 50
    for q, w in zip(qs, want):
        assert int(qi.and_queries([q])[0]) == w
    assert list(qi.and_queries(qs[:2] + [[3, 0]])) == want[:2] + [intersect(docids, bounds, [3, 0])]  # a small batch
    fdev = device.Dictionary(kind, fd)
    c, sm, _ = qi.and_queries_with_freqs(fdev, [[2, 0, 1]])
    assert int(c[0]) == want[2] and int(sm[0]) == 3 * want[2]  # every freq is 1
    qi.close()


@pytest.mark.parametrize("lean_pages", ["0", "1000000000"])
@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_blocks_left_as_gaps(device, kind, lean_pages):
    """Blocks the expansion cannot turn into docIDs on the fly — a dictionary entry holding a value >= 65536 (a
    constant stride of 70000: the dictionary learns runs of 69999), or more than 256 slots in the block (every gap
    an exception) — are summed afterwards, by the decoding wave or by the fix-up launch."""
    device.set_option("query_lean_pages", int(lean_pages))
    r = np.random.default_rng(77)
    stride = (np.arange(700, dtype=np.uint64) * 70000).astype(np.uint32)
    wild = np.cumsum(r.integers(100000, 3000000, 700, dtype=np.uint64)).astype(np.uint32)
    both = np.union1d(stride, wild).astype(np.uint32)
    dense = np.arange(0, 2_000_000, 7, dtype=np.uint32)
    lists = [stride, wild, both, dense]
    docids = np.concatenate(lists)
    lens = np.array([len(x) for x in lists], dtype=np.uint32)
    bounds = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    gaps = np.concatenate([host.docids_to_gaps(x) for x in lists])
    coll = host.Collection(gaps, lens)
    freqs = np.ones(docids.size, dtype=np.uint32)
    dd = host.build_dictionary(kind, coll)
    fd = host.build_dictionary(kind, host.Collection(freqs - 1, lens))
    idx, offs = host.build_index(kind, dd, fd, docids, freqs, lens)
    qi = device.QueryIndex(device.Dictionary(kind, dd), idx, offs)
    qs = [[0, 2], [1, 2], [0, 1], [0, 3], [1, 3], [2, 3], [0, 1, 2], [3, 2, 1]]
    want = [intersect(docids, bounds, q) for q in qs]
    assert want[0] == len(stride) and want[1] == len(wild)
    assert list(qi.and_queries(qs)) == want
    for q, w in zip(qs, want):
        assert int(qi.and_queries([q])[0]) == w
    od = oracle.OracleDict(kind, dd)
    assert oracle.and_query(od, idx, offs, int(docids.max()) + 1, [0, 2]) == want[0]
    qi.close()


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.RECTANGULAR, host.MULTI_PACKED])
@pytest.mark.parametrize("corpus_name", ["small_corpus", "dense_corpus"])
def test_with_freqs_matches_oracle(device, request, kind, corpus_name):
    """and_query<true> (queries.hpp:72-76): the freqs of every term at every match, freqs parts decoded lazily."""
    corpus = request.getfixturevalue(corpus_name)
    ix = get_index(corpus, kind)
    dd = device.Dictionary(kind, ix.docs_dict)
    fd = device.Dictionary(kind, ix.freqs_dict)
    qi = device.QueryIndex(dd, ix.bytes, ix.offsets)
    qs = reference_queries(len(ix.lens))[:150] + heavy_queries(ix.lens, 120, seed=11)
    counts, sums, nblocks = qi.and_queries_with_freqs(fd, qs)
    want = [intersect_freqs(ix.docids, ix.freqs, ix.bounds, q) for q in qs]
    assert np.array_equal(counts, np.array([w[0] for w in want], dtype=np.uint64))
    assert np.array_equal(sums, np.array([w[1] for w in want], dtype=np.uint64))
    assert np.array_equal(counts, qi.and_queries(qs))
    assert int(sums.sum()) > int(counts.sum()) > 1000
    # one query at a time: the same freqs blocks as the reference's lazy freq() decodes
    oi = oracle.OracleIndex(oracle.OracleDict(kind, ix.docs_dict), ix.bytes, ix.offsets, int(ix.docids.max()) + 1)
    ofd = oracle.OracleDict(kind, ix.freqs_dict)
    total_blocks = len(qi.blocks)
    lazy = 0
    for i in range(0, len(qs), 9):
        c1, s1, b1 = qi.and_queries_with_freqs(fd, [qs[i]])
        assert (int(c1[0]), int(s1[0]), b1) == oi.and_query_freqs(ofd, qs[i])
        n_terms = np.unique(qs[i]).size
        lazy += b1 < sum(int(np.ceil(ix.lens[t] / 256)) for t in np.unique(qs[i]))
    assert lazy > 0 and total_blocks > 0  # fewer freqs parts than the lists hold, for some queries at least
    qi.close()


def test_with_freqs_edges(device, dense_corpus):
    kind = host.SINGLE_PACKED
    ix = get_index(dense_corpus, kind)
    dd = device.Dictionary(kind, ix.docs_dict)
    fd = device.Dictionary(kind, ix.freqs_dict)
    qi = device.QueryIndex(dd, ix.bytes, ix.offsets)
    c, s, b = qi.and_queries_with_freqs(fd, [])
    assert c.size == 0 and s.size == 0 and b == 0
    big = int(np.argmax(ix.lens))
    c, s, b = qi.and_queries_with_freqs(fd, [[big], [big, big], []])
    lo, hi = int(ix.bounds[big]), int(ix.bounds[big + 1])
    assert list(c) == [hi - lo, hi - lo, 0]
    assert list(s) == [int(ix.freqs[lo:hi].astype(np.uint64).sum())] * 2 + [0]
    wrong = device.Dictionary(host.RECTANGULAR, get_index(dense_corpus, host.RECTANGULAR).freqs_dict)
    with pytest.raises(device.DintError):
        qi.and_queries_with_freqs(wrong, [[big]])
    qi.close()


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_reference_query_log_on_the_readme_shaped_collection(device, kind):
    """BASELINE config 5 on config 1's collection: the reference's query log, term ids as they are, over an index of the
    shape README.md:53 gives (113 306 lists) — batched and one query per call (op_perftest, src/queries.cpp:15-61),
    with and without freqs, against the oracle and plain set intersection."""
    ix = ReadmeIndex(kind)
    qs = reference_queries(len(ix.lens))
    dd = device.Dictionary(kind, ix.docs_dict)
    fd = device.Dictionary(kind, ix.freqs_dict)
    qi = device.QueryIndex(dd, ix.bytes, ix.offsets)
    want = np.array([intersect(ix.docids, ix.bounds, q) for q in qs], dtype=np.uint64)
    assert np.array_equal(qi.and_queries(qs), want) and int(want.sum()) > 500
    counts, sums, _ = qi.and_queries_with_freqs(fd, qs)
    wf = [intersect_freqs(ix.docids, ix.freqs, ix.bounds, q) for q in qs]
    assert np.array_equal(counts, want) and np.array_equal(sums, np.array([w[1] for w in wf], dtype=np.uint64))
    oi = oracle.OracleIndex(oracle.OracleDict(kind, ix.docs_dict), ix.bytes, ix.offsets, 10_000)
    for i in range(0, len(qs), 11):
        assert int(qi.and_queries([qs[i]])[0]) == oi.and_query(qs[i]) == int(want[i])
    qi.close()


def test_one_query_index_under_two_host_threads(device, small_corpus):
    """SURVEY §8b "Threading": one dint_query_index, two host threads, a stream each, calls of every form interleaved for a
    few seconds — single light queries (one launch), single heavy ones (a round per launch), batches of small queries (a
    workgroup per query), mixed batches (split calls), and_query<true> calls — every result against plain set intersection.
    The calls of one handle serialise on the handle's own lock (include/dint_hip.h); nothing else is shared."""
    import threading
    import time

    import torch

    kind = host.SINGLE_PACKED
    ix = get_index(small_corpus, kind)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    qi = device.QueryIndex(dd, ix.bytes, ix.offsets)
    light = reference_queries(len(ix.lens))[:200]
    heavy = heavy_queries(ix.lens, 40, seed=11)
    want_light = np.array([intersect(ix.docids, ix.bounds, q) for q in light], dtype=np.uint64)
    want_heavy = np.array([intersect(ix.docids, ix.bounds, q) for q in heavy], dtype=np.uint64)
    want_fsum = [intersect_freqs(ix.docids, ix.freqs, ix.bounds, q) for q in heavy[:12]]
    errors, calls = [], [0, 0]
    stop_at = time.monotonic() + 4.0

    def worker(k):
        try:
            stream = torch.cuda.Stream(torch.device("cuda", 0))
            r = np.random.default_rng(100 + k)
            terms_l, offs_l = device._pack_queries(light)
            terms_m, offs_m = device._pack_queries(light[:60] + heavy)
            while time.monotonic() < stop_at:
                form = int(r.integers(0, 5))
                if form == 0:    # one light query
                    i = int(r.integers(0, len(light)))
                    t, o = device._pack_queries([light[i]])
                    c = np.zeros(1, dtype=np.uint64)
                    qi.and_queries_packed(t, o, c, stream.cuda_stream)
                    assert int(c[0]) == int(want_light[i]), ("light", i)
                elif form == 1:  # one heavy query
                    i = int(r.integers(0, len(heavy)))
                    t, o = device._pack_queries([heavy[i]])
                    c = np.zeros(1, dtype=np.uint64)
                    qi.and_queries_packed(t, o, c, stream.cuda_stream)
                    assert int(c[0]) == int(want_heavy[i]), ("heavy", i)
                elif form == 2:  # the light log as one batch
                    c = np.zeros(len(light), dtype=np.uint64)
                    qi.and_queries_packed(terms_l, offs_l, c, stream.cuda_stream)
                    assert np.array_equal(c, want_light), "batch"
                elif form == 3:  # a mixed batch
                    c = np.zeros(60 + len(heavy), dtype=np.uint64)
                    qi.and_queries_packed(terms_m, offs_m, c, stream.cuda_stream)
                    assert np.array_equal(c, np.r_[want_light[:60], want_heavy]), "mixed"
                else:            # and_query<true>
                    with torch.cuda.stream(stream):
                        counts, sums, _ = qi.and_queries_with_freqs(fd, heavy[:12])
                    assert [(int(a), int(b)) for a, b in zip(counts, sums)] == want_fsum, "freqs"
                calls[k] += 1
        except BaseException as e:  # noqa: BLE001 (reported by the main thread)
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert min(calls) >= 20, calls
    qi.close()

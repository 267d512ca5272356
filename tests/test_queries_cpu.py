"""AND queries on the CPU: the oracle's and_query (block-max skipping cursors, a restatement of
include/ds2i/queries.hpp:34-84 + dict_posting_list.hpp:111-147) against plain set intersection
of the index builder's input."""
import numpy as np
import pytest

import oracle
from dint_amd import host
from queries import ReadmeIndex, heavy_queries, intersect, intersect_freqs, reference_queries
from test_index_cpu import get_index


def _num_docs(ix):
    return int(ix.docids.max()) + 1


def test_reference_query_log_shape():
    qs = reference_queries(1 << 30)
    assert len(qs) == 500
    assert min(len(q) for q in qs) == 1 and max(len(q) for q in qs) == 11


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.RECTANGULAR, host.MULTI_PACKED])
def test_oracle_and_query_is_set_intersection(small_corpus, kind):
    ix = get_index(small_corpus, kind)
    od = oracle.OracleDict(kind, ix.docs_dict)
    nd = _num_docs(ix)
    qs = reference_queries(len(ix.lens))[:150] + heavy_queries(ix.lens, 60)
    hits = 0
    for q in qs:
        got = oracle.and_query(od, ix.bytes, ix.offsets, nd, q)
        assert got == intersect(ix.docids, ix.bounds, q)
        hits += got
    assert hits > 1000  # the workload does exercise non-empty intersections


def test_oracle_and_query_edges(dense_corpus):
    kind = host.SINGLE_PACKED
    ix = get_index(dense_corpus, kind)
    od = oracle.OracleDict(kind, ix.docs_dict)
    nd = _num_docs(ix)
    longest = int(np.argmax(ix.lens))
    shortest = int(np.argmin(ix.lens))
    assert oracle.and_query(od, ix.bytes, ix.offsets, nd, []) == 0  # queries.hpp:38
    assert oracle.and_query(od, ix.bytes, ix.offsets, nd, [longest]) == int(ix.lens[longest])
    assert oracle.and_query(od, ix.bytes, ix.offsets, nd, [longest] * 4) == int(ix.lens[longest])  # :28-31
    assert oracle.and_query(od, ix.bytes, ix.offsets, nd, [shortest, longest]) == intersect(
        ix.docids, ix.bounds, [shortest, longest])


def test_reference_query_log_on_the_readme_shaped_collection():
    """the reference's own query log (test/test_data/queries) over an index of the
    shape its README gives for the test collection — term ids as they are, no folding. and_query<false> and <true> of the
    oracle against plain set intersection of the builder's input."""
    kind = host.SINGLE_PACKED
    ix = ReadmeIndex(kind)
    qs = reference_queries(len(ix.lens))
    assert len(ix.lens) == 113_306 and max(int(q.max()) for q in qs) == 113_242
    oi = oracle.OracleIndex(oracle.OracleDict(kind, ix.docs_dict), ix.bytes, ix.offsets, 10_000)
    ofd = oracle.OracleDict(kind, ix.freqs_dict)
    hits = 0
    for i, q in enumerate(qs):
        want = intersect(ix.docids, ix.bounds, q)
        assert oi.and_query(q) == want
        if i % 5 == 0:
            n, fsum, _ = oi.and_query_freqs(ofd, q)
            assert (n, fsum) == intersect_freqs(ix.docids, ix.freqs, ix.bounds, q)
        hits += want
    assert hits > 500


def test_query_log_on_several_threads_equals_one_by_one(small_corpus):
    """oracle_and_queries_parallel (pthreads inside liboracle, query q on thread q % threads: the all-cores CPU figure of
    tests/query_timing.py): the same counts as one query after the other, for any thread count, empty queries included."""
    kind = host.SINGLE_PACKED
    ix = get_index(small_corpus, kind)
    oi = oracle.OracleIndex(oracle.OracleDict(kind, ix.docs_dict), ix.bytes, ix.offsets, _num_docs(ix))
    qs = reference_queries(len(ix.lens))[:120] + heavy_queries(ix.lens, 40) + [np.zeros(0, dtype=np.uint32)]
    want = np.array([oi.and_query(q) for q in qs], dtype=np.uint64)
    for threads in (1, 3, 8):
        got, wall = oi.and_queries_parallel(qs, threads, passes=2)
        assert np.array_equal(got, want) and wall > 0

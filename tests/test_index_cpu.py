"""In-index path on the CPU: interpolative coding, dict_posting_list layout, oracle walker."""
import numpy as np
import pytest

import oracle
from dint_amd import host


class Index:
    def __init__(self, corpus, kind, seed=3):
        coll = corpus.coll
        self.docids = host.gaps_to_docids(coll)
        self.freqs = host.synth_freqs(coll.num_postings, seed)
        self.lens = coll.lens
        self.bounds = coll.list_bounds()
        self.docs_dict = corpus.dict_file(kind)
        self.freqs_dict = host.build_dictionary(kind, host.Collection(self.freqs - 1, coll.lens))
        self.bytes, self.offsets = host.build_index(kind, self.docs_dict, self.freqs_dict, self.docids, self.freqs,
                                                     coll.lens)


_cache = {}


def get_index(corpus, kind):
    key = (id(corpus), kind)
    if key not in _cache:
        _cache[key] = Index(corpus, kind)
    return _cache[key]


def test_gaps_docids_inverse(small_corpus):
    coll = small_corpus.coll
    docids = host.gaps_to_docids(coll)
    b = coll.list_bounds()
    for i in range(0, len(coll.lens), 37):
        lo, hi = int(b[i]), int(b[i + 1])
        assert np.array_equal(host.docids_to_gaps(docids[lo:hi]), coll.gaps[lo:hi])
        assert (np.diff(docids[lo:hi].astype(np.int64)) > 0).all()


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 100, 255])
def test_interpolative_round_trip(n):
    """interpolative_block::encode -> decode, with the sum given and with sum_of_values = -1
    (test/test_block_codecs.cpp:9-38 shape: seeded values, consumed bytes == produced bytes)."""
    r = np.random.default_rng(n)
    for mag in (1, 5, 12, 20):
        vals = r.integers(0, 1 << mag, n, dtype=np.uint64).astype(np.uint32)
        docs = (np.cumsum(vals.astype(np.uint64) + 1) - 1).astype(np.uint32)
        freqs = vals + 1
        # a one-list index with n < 256 is exactly: vbyte(n) | max | docs interpolative | freqs interpolative
        d = host.build_dictionary(host.SINGLE_PACKED, host.Collection(vals, np.array([n], dtype=np.uint32)))
        idx, offs = host.build_index(host.SINGLE_PACKED, d, d, docs, freqs, np.array([n], dtype=np.uint32))
        hdr = 1 if n < 128 else 2
        assert int.from_bytes(bytes(idx[hdr:hdr + 4]), "little") == int(docs[-1])
        got, used = oracle.interpolative_decode(idx, hdr + 4, int(docs[-1]) - (n - 1), n)
        assert np.array_equal(got, vals)
        got_f, used_f = oracle.interpolative_decode(idx, hdr + 4 + used, 0xFFFFFFFF, n)
        assert np.array_equal(got_f, vals)
        assert hdr + 4 + used + used_f == idx.size


@pytest.mark.parametrize("kind", [host.RECTANGULAR, host.SINGLE_PACKED, host.MULTI_PACKED])
def test_index_round_trip(small_corpus, kind):
    ix = get_index(small_corpus, kind)
    od, of = oracle.OracleDict(kind, ix.docs_dict), oracle.OracleDict(kind, ix.freqs_dict)
    assert ix.offsets[-1] == ix.bytes.size
    for i in range(len(ix.lens)):
        if ix.lens[i] == 0:
            continue
        docids, freqs = oracle.posting_list_decode(od, of, ix.bytes, int(ix.offsets[i]))
        lo, hi = int(ix.bounds[i]), int(ix.bounds[i + 1])
        assert np.array_equal(docids, ix.docids[lo:hi])
        assert np.array_equal(freqs, ix.freqs[lo:hi])


def test_full_blocks_are_vroom_segments(small_corpus):
    """A full in-index block is byte-identical to the whole-list coding of the same 256 integers."""
    coll = small_corpus.coll
    kind = host.SINGLE_PACKED
    ix = get_index(small_corpus, kind)
    i = int(np.argmax(coll.lens))
    lo = int(coll.list_bounds()[i])
    first_block = host.Collection(coll.gaps[lo:lo + 256].copy(), np.array([256], dtype=np.uint32))
    enc, units = host.encode_vroom(kind, ix.docs_dict, first_block, unit_ints=0)
    payload = bytes(enc[int(units["in_off"][0]):])
    n_blocks = (int(coll.lens[i]) + 255) // 256
    n = int(coll.lens[i])
    hdr = 1 if n < 128 else (2 if n < 16384 else 3)
    start = int(ix.offsets[i]) + hdr + 4 * n_blocks + 4 * (n_blocks - 1)
    assert bytes(ix.bytes[start:start + len(payload)]) == payload


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_oracle_and_query_with_freqs(small_corpus, kind):
    """and_query<true> (queries.hpp:72-76) against plain set intersection of the builder's input."""
    from queries import heavy_queries, intersect_freqs, reference_queries

    ix = get_index(small_corpus, kind)
    oi = oracle.OracleIndex(oracle.OracleDict(kind, ix.docs_dict), ix.bytes, ix.offsets, int(ix.docids.max()) + 1)
    ofd = oracle.OracleDict(kind, ix.freqs_dict)
    qs = reference_queries(len(ix.lens))[:60] + heavy_queries(ix.lens, 40, seed=2)
    seen = 0
    for q in qs:
        n, fsum, blocks = oi.and_query_freqs(ofd, q)
        assert (n, fsum) == intersect_freqs(ix.docids, ix.freqs, ix.bounds, q)
        assert n == oi.and_query(q)
        assert (blocks == 0) == (n == 0)
        seen += n
    assert seen > 100

"""Query workloads for the AND-query tests.

tests/golden/queries.txt is the query log the reference's own tests ship
(test/test_data/queries: 500 queries of 1-11 term ids, one per line); its
term ids address the reference's test collection, so they are folded onto
the synthetic corpora with `term % n_lists`."""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_queries(n_lists: int):
    """n_lists >= 113 243 (the log's largest term id + 1, e.g. host.readme_test_collection): the log as it is."""
    out = []
    with open(os.path.join(_HERE, "golden", "queries.txt")) as f:
        for line in f:
            terms = [int(t) % n_lists for t in line.split()]
            if terms:
                out.append(np.array(terms, dtype=np.uint32))
    return out


def heavy_queries(lens: np.ndarray, n_queries: int, seed: int = 5, pool: int = 48, max_terms: int = 6):
    """Queries over the longest lists, so that intersections are non-empty and span many blocks."""
    r = np.random.default_rng(seed)
    big = np.argsort(-lens.astype(np.int64), kind="stable")[:pool]
    out = []
    for _ in range(n_queries):
        k = int(r.integers(1, max_terms + 1))
        out.append(r.choice(big, k, replace=True).astype(np.uint32))  # duplicates on purpose
    return out


def intersect(docids: np.ndarray, bounds: np.ndarray, terms) -> int:
    """Plain set intersection of the lists' docIDs (the builder's input, no codec involved)."""
    terms = np.unique(np.asarray(terms))
    if terms.size == 0:
        return 0
    cur = docids[int(bounds[terms[0]]):int(bounds[terms[0] + 1])]
    for t in terms[1:]:
        cur = np.intersect1d(cur, docids[int(bounds[t]):int(bounds[t + 1])], assume_unique=True)
    return int(cur.size)


def intersect_freqs(docids: np.ndarray, freqs: np.ndarray, bounds: np.ndarray, terms):
    """-> (matches, sum over matches and over the distinct terms of the term's freq in the document)."""
    terms = np.unique(np.asarray(terms))
    if terms.size == 0:
        return 0, 0
    cur = docids[int(bounds[terms[0]]):int(bounds[terms[0] + 1])]
    for t in terms[1:]:
        cur = np.intersect1d(cur, docids[int(bounds[t]):int(bounds[t + 1])], assume_unique=True)
    total = 0
    for t in terms:
        lo, hi = int(bounds[t]), int(bounds[t + 1])
        pos = np.searchsorted(docids[lo:hi], cur)
        total += int(freqs[lo:hi][pos].astype(np.uint64).sum())
    return int(cur.size), total


class ReadmeIndex:
    """The README-shaped stand-in for the reference's test collection (host.readme_test_collection) as an index:
    113 306 lists, so the reference's query log (term ids up to 113 242) addresses it without folding."""
    _cache = {}

    def __new__(cls, kind):
        if kind not in cls._cache:
            from dint_amd import host

            self = object.__new__(cls)
            coll = host.readme_test_collection(seed=1)
            self.docids = host.gaps_to_docids(coll)
            self.freqs = host.synth_freqs(coll.num_postings, 3)
            self.lens = coll.lens
            self.bounds = coll.list_bounds()
            self.docs_dict = host.build_dictionary(kind, coll)
            self.freqs_dict = host.build_dictionary(kind, host.Collection(self.freqs - 1, coll.lens))
            self.bytes, self.offsets = host.build_index(kind, self.docs_dict, self.freqs_dict, self.docids, self.freqs, coll.lens)
            cls._cache[kind] = self
        return cls._cache[kind]

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The CPU tests need the offline host library and the oracle (gcc / g++ only); the HIP library is built too
    when hipcc is there (tests/test_abi.py loads it: no GPU needed for that) — normally __graft_entry__.build()
    has made all three already."""
    import shutil
    import subprocess

    csrc = os.path.join(ROOT, "dint_amd", "csrc")
    host_so = os.path.join(ROOT, "dint_amd", "libdint_host.so")
    hip_so = os.path.join(ROOT, "dint_amd", "libdint_hip.so")
    if not os.path.exists(host_so):
        subprocess.run(["make", "-C", csrc, host_so], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hip_so) and os.path.exists(hipcc):
        subprocess.run(["make", "-C", csrc, hip_so], check=True)


_ensure_built()

from dint_amd import host  # noqa: E402


class Corpus:
    """A small synthetic collection with its three dictionaries and encoded streams."""

    def __init__(self, postings, universe, seed, unit_ints=1024, **params):
        self.coll = host.synth_collection(postings, universe=universe, seed=seed, **params)
        self.unit_ints = unit_ints
        self._dicts = {}
        self._enc = {}

    def dict_file(self, kind):
        if kind not in self._dicts:
            self._dicts[kind] = host.build_dictionary(kind, self.coll)
        return self._dicts[kind]

    def encoded(self, kind, greedy=False):
        key = (kind, greedy)
        if key not in self._enc:
            self._enc[key] = host.encode_vroom(kind, self.dict_file(kind), self.coll, unit_ints=self.unit_ints,
                                               greedy=greedy)
        return self._enc[key]


@pytest.fixture(scope="session")
def small_corpus():
    # ~400k postings, lists from 1 to ~66k long: every size class, runs and exceptions occur
    return Corpus(400_000, universe=200_000, seed=7)


@pytest.fixture(scope="session")
def dense_corpus():
    # few long dense lists: long zero runs, 16-entries
    return Corpus(300_000, universe=120_000, seed=11, min_len=20_000, p_cluster_max=0.995)


@pytest.fixture(scope="session")
def sparse_corpus():
    # many short sparse lists: exceptions dominate
    return Corpus(120_000, universe=50_000_000, seed=3, max_len=300)


def rng(seed):
    return np.random.default_rng(seed)

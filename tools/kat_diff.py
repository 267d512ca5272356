#!/usr/bin/env python3
"""Development aid (GPU box): where the device's decode of the hand-assembled known-answer streams differs from the expected integers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from dint_amd import device
from kat import DICT_FILES, cases
for kind in (1, 0):
    d = device.Dictionary(kind, DICT_FILES[kind])
    for name, buf, off, n, expect in cases("single_cases"):
        got, consumed = d.decode_list(buf, off, n)
        bad = np.nonzero(got != expect)[0]
        if bad.size:
            print(f"kind {kind} {name}: n {n}, stream {buf.size - off} B; {bad.size} wrong, first at {bad[0]}, last at {bad[-1]}; got {got[bad[0]:bad[0]+6]} want {expect[bad[0]:bad[0]+6]}")
print("done")

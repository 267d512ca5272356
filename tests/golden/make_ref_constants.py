#!/usr/bin/env python3
"""Generates tests/golden/ref_constants.json from the REFERENCE's own dint_configuration.hpp.

Needs oracle/_ref/libref_constants.so, i.e. /root/reference must be present: `make -C oracle ref` compiles
oracle/ref_constants_check.cpp against /root/reference/include/dint/dint_configuration.hpp where it lies (the header
needs <cmath> and <limits> only) — its static_asserts against dint/constants.hpp hold if that build succeeds. The values
(data only) travel; the reference does not."""
import ctypes as C
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libref_constants.so"))
assert lib.ref_constants_check() == 0
buf = (C.c_uint32 * 32)()
n = lib.ref_constants(buf, 32)
names = ["EXCEPTIONS", "num_selectors", "max_entry_size", "num_entries", "log2_num_entries", "num_target_sizes",
         "target_sizes[0]", "target_sizes[1]", "target_sizes[2]", "target_sizes[3]", "target_sizes[4]"]
assert n == len(names)
with open(os.path.join(HERE, "ref_constants.json"), "w") as f:
    json.dump({"about": "values of /root/reference/include/dint/dint_configuration.hpp:6,20,24-28 as compiled by g++",
               "constants": {k: int(buf[i]) for i, k in enumerate(names)}}, f, indent=1)
print({k: int(buf[i]) for i, k in enumerate(names)})

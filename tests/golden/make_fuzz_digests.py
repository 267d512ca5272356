#!/usr/bin/env python3
"""Generates tests/golden/fuzz_digests.json: for every case of the committed fuzz plans (tests/fuzz_streams.py: the seeds
and shapes ARE the plan) a digest of the dictionary file, the stream, its unit table and the integers the generator says it
decodes to — written down by substitution, no decoder runs here. The tests rebuild every case from its seed, compare the
digest (the generator is deterministic and has not drifted), then decode."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fuzz_streams as F  # noqa: E402

VROOM = (24, 400)   # dictionaries per kind, lists per dictionary: 3 x 24 x 400 = 28 800 lists
INDEX = (10, 200)   # 3 x 10 x 200 = 6 000 posting lists

if __name__ == "__main__":
    out = {"vroom_plan": list(VROOM), "index_plan": list(INDEX), "vroom": {}, "index": {}}
    for case in F.plan(*VROOM):
        D, S = F.build_case(case)
        out["vroom"][str(case[0])] = {"digest": F.digest(D, S), "lists": len(S.lists), "units": len(S.units),
                                      "integers": int(len(S.expect)), "bytes": int(len(S.enc))}
    for case in F.index_plan(*INDEX):
        Dd, Df, X = F.build_index_case(case)
        out["index"][str(case[0])] = {"digest": F.index_digest(Dd, Df, X), "lists": len(X.offsets) - 1,
                                      "postings": int(len(X.docids)), "bytes": int(len(X.index))}
    with open(os.path.join(HERE, "fuzz_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(sum(v["lists"] for v in out["vroom"].values()), "vroom lists,", sum(v["lists"] for v in out["index"].values()),
          "posting lists")

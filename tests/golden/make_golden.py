#!/usr/bin/env python3
"""Generate the golden vectors by running the REFERENCE runtime (oracle/_ref/ocr_replay_dbg, built by
oracle/Makefile.ref from /root/reference) on every test stream: 1 worker, COUNTED_MAP provider, and a
4-worker cross-check.  Fixtures are data only: the op stream (inputs) and the result record (outputs);
large streams keep a digest + per-level hashes instead of the arrays.  Run from the repo root."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import streams  # noqa: E402
from oracle import ref as O  # noqa: E402

O.build()
index = {}
for s in streams.all_streams():
    r1, info = O.run_reference(s, workers=1)
    r4, _ = O.run_reference(s, workers=4)
    assert r1.diff(r4) == [], (s.name, r1.diff(r4))
    entry = dict(n_edt=s.n_edt, n_evt=s.n_evt, n_db=s.n_db, n_fired=r1.n_fired, digest=f"{r1.digest():016x}",
                 levels={str(k): [v[0], f"{v[1]:016x}"] for k, v in r1.level_hashes().items()})
    if s.n_edt <= 2000:
        s.to_file(os.path.join(HERE, s.name + ".ops"))
        r1.to_file(os.path.join(HERE, s.name + ".res"))
        entry["files"] = True
    index[s.name] = entry
    print(s.name, entry["n_fired"], entry["digest"])
with open(os.path.join(HERE, "index.json"), "w") as f:
    json.dump(index, f, indent=1, sort_keys=True)

# Fuzz corpus: random well-formed OCR programs (streams.random_stream); the reference's result digest per seed, both
# generator flavours, checked at 1 and 4 workers.  The streams are regenerated from the seed by the tests; a drift of
# the generator shows as a digest of the op arrays that no longer matches.
fuzz = {}
for safe in (False, True):
    for seed in range(streams.FUZZ_SEEDS):
        s = streams.random_stream(seed, eager_safe=safe)
        r1, _ = O.run_reference(s, workers=1)
        r4, _ = O.run_reference(s, workers=4)
        assert r1.diff(r4) == [], (s.name, r1.diff(r4))
        fuzz[s.name] = dict(input=f"{streams.stream_digest(s):016x}", n_fired=r1.n_fired, digest=f"{r1.digest():016x}")
with open(os.path.join(HERE, "fuzz.json"), "w") as f:
    json.dump(fuzz, f, indent=0, sort_keys=True)
print("fuzz:", len(fuzz), "programs")

#!/usr/bin/env python3
"""Generate tests/golden/lockable.json: what the REFERENCE runtime (oracle/_ref, cfg datablockfactory=Lockable) did on
the streams of tests/lockable.py with 1 and with 4 workers -- the levels its arbitration produced (one valid outcome
each), the digest of everything else in the record, and the number of overlapping holders its bodies saw (0).
Run in the container that has /root/reference:  python tests/golden/make_lockable_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]

import lockable as L  # noqa: E402
from oracle import ref as O  # noqa: E402


def rest_digest(r):
    import hashlib
    h = hashlib.sha256()
    for a in (r.val, r.evt_state, r.db_val):
        h.update(a.tobytes())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    out = {}
    for s in L.all_lockable():
        ent = {}
        for w in (1, 4):
            r, info = O.run_reference(s, workers=w, db="Lockable")
            ent[f"workers{w}"] = dict(level=r.level.tolist(), rest=rest_digest(r), overlaps=info["overlaps"], n_fired=int(r.n_fired))
        r, info = O.run_reference(s, workers=4, db="Regular", lock_harness=True)   # what the watcher sees without arbitration
        ent["regular_workers4_overlaps"] = info["overlaps"]
        out[s.name] = ent
        print(s.name, ent["workers1"]["overlaps"], ent["workers4"]["overlaps"], ent["regular_workers4_overlaps"])
    with open(os.path.join(HERE, "lockable.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)

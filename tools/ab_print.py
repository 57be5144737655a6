#!/usr/bin/env python3
"""Print the bench lines of an A/B series: python tools/ab_print.py pre1 new1 ... (reads gpurun_out/ab_<name>.json)."""
import json
import sys

for n in sys.argv[1:]:
    j = json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
    print(n, "sat/s %.3fe9" % (j["value"] / 1e9), "ms/step %.3f" % j["ms_per_step"], "us/launch %.3f" % j["roofline"]["us_per_launch"],
          "launches %d" % j["roofline"]["launches_per_resolve"], "parity", j["config"]["parity"])

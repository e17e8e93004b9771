#!/usr/bin/env python3
"""stdin: one bench.py JSON line; prints `<label> value (min max) ms_per_step` -- for shell loops over bench settings."""
import json
import sys

d = json.loads(sys.stdin.read())
print(" ".join(sys.argv[1:]), round(d["value"]), "(%d %d)" % (round(d.get("value_min", 0)), round(d.get("value_max", 0))), "%.4f ms" % d["ms_per_step"])

"""stdin: bench.py output; prints tok/s, per-kernel launch durations and the roofline fraction of the JSON line (A/B sweeps)"""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        j = json.loads(line)
        print(round(j["value"], 1), "tok/s", {s["kernel"]: s["us"] for s in j.get("mm8_one_shapes", [])}, "frac", j.get("roofline", {}).get("frac"))

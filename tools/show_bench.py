"""Print the summary block of a bench.py JSON line (file argument, or stdin)."""
import json
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads([ln for ln in txt.strip().splitlines() if ln.startswith("{")][-1])
for k, v in d.get("summary", {"value": d.get("value"), "ms_per_step": d.get("ms_per_step")}).items():
    print(f"{k:32s} {v}")

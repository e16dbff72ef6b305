#!/bin/bash
# A/B of an environment switch on one GPU box, interleaved rounds (each run is its own process: the switches are read once
# per process):   tools/ab_env.sh <VAR> "<v1> <v2> .." [rounds] [presets]
#   e.g.  tools/ab_env.sh VITSMI_XCD_GROUP "1 4 8" 3 "high medium"
# prints one line per run: preset, VAR=value, samples/s, ms per step.
VAR=$1; VALS=$2; ROUNDS=${3:-3}; PRESETS=${4:-"high medium"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in $(seq 1 $ROUNDS); do
  for p in $PRESETS; do
    parts=2; [ "$p" = medium ] && parts=3
    for v in $VALS; do
      out=$(env $VAR=$v python3 $R/bench.py --preset $p --parts $parts --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
      echo "$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('round $r  %-7s $VAR=%-4s %10.2f M samples/s  %8.3f ms/step' % ('$p', '$v', d['value']/1e6, d['ms_per_step']))"
    done
  done
done

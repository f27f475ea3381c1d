#!/bin/bash
# usage: ab.sh "<label>:<env assignments>" ...   -> ms_per_step per variant, two rounds
for round in 1 2; do
  for v in "$@"; do
    label=${v%%:*}; envs=${v#*:}
    out=$(env $envs timeout -k 10 150 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
    echo "$label $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["probe"]["frac"])')"
  done
done

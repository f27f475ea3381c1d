#!/bin/bash
# usage: [BENCH_ARGS="--network Swin34"] [ROUNDS=2] ab_env.sh "<label>:<env assignments>" ...   -> ms_per_step (+ roofline frac, probe frac) per variant,
# alternating on ONE box
for round in $(seq ${ROUNDS:-2}); do
  for v in "$@"; do
    label=${v%%:*}; envs=${v#*:}
    out=$(env $envs timeout -k 10 150 python bench.py --no-cpu-baseline --no-extra ${BENCH_ARGS:-} 2>/dev/null | tail -1)
    echo "$label $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["probe"]["frac"])')"
  done
done

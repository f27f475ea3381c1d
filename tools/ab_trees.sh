# step-level A/B of two TREES (the working tree against a git worktree under ./_old): bash tools/ab_trees.sh <passes> <bench args...>
passes=$1; shift
for pass in $(seq $passes); do
  for t in _old .; do
    echo "== tree $t"; (cd $t && python bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  done
done

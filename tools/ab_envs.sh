#!/bin/bash
# several environment settings against each other on ONE box, alternating: tools/ab_envs.sh "<bench args>" "VAR=a" "VAR=b OTHER=c" ...   ("-" = no setting)
ARGS=$1; shift
for r in 1 2 3; do
  for kv in "$@"; do
    [ "$kv" == "-" ] && E="" || E="$kv"
    env $E python bench.py --steps 20 --warmup 3 --no-cpu-baseline $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % sys.argv[1], d['ms_per_step'], d['value'])" "$kv"
  done
done

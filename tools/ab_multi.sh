#!/bin/bash
# several environment variants against the default on ONE box: tools/ab_multi.sh "VAR=.. VAR2=.." "VAR=.." ...
run() { env $1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-44s' % sys.argv[1], d['ms_per_step'], d['value'])" "$1"; }
for r in 1 2; do
  run "A=default"
  for kv in "$@"; do run "$kv"; done
done

#!/bin/bash
# several experiment libraries / env switches against the default on ONE box, two rounds:
#   tools/ab_multi2.sh "ENV=VAL" "OFDFT_LIB=build_ab/lib_x.so" ...
run() { env $1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-44s' % sys.argv[1], d['ms_per_step'], d['value'], {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.018})" "$1"; }
for r in 1 2; do
  run "A=default"
  for v in "$@"; do run "$v"; done
done

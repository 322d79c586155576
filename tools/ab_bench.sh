#!/bin/bash
# A/B on ONE box: alternate the default library and an experiment build (OFDFT_LIB), 3 rounds each
# usage: tools/ab_bench.sh <experiment .so> [bench args]
EXP=$1; shift
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A default  ', d['ms_per_step'], {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.03})"
  OFDFT_LIB=$EXP python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B experiment', d['ms_per_step'], {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.03})"
done

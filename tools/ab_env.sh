#!/bin/bash
# A/B on ONE box by environment variable: tools/ab_env.sh VAR=VALUE [bench args]
KV=$1; shift
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A default', d['ms_per_step'], d['value'], repr(d['energy_Ha']), {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.04})"
  env $KV python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B $KV', d['ms_per_step'], d['value'], repr(d['energy_Ha']), {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.04})"
done

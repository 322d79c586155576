#!/bin/bash
# as tools/mixed_probe.sh, fused pipelines only (no chirp-z comparison).   usage: bash tools/mixed_probe_fast.sh [sizes...]
run() { python bench.py --grid $1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); n=d['config']['grid'][0]
print(json.dumps({'grid': n, 'ms_per_eval': d['ms_per_step'], 'evals_per_s': d['value'], 'ns_per_point': round(d['ms_per_step']*1e6/n**3, 4),
                  'kernels': {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.04}}))"; }
for n in ${@:-48 96 120 144 160 192 240 250 270 288 320 384 480}; do run $n; done

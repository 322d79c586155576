#!/bin/bash
# evaluations/s of the cfg3 closure on grids with factors 3 and 5: mixed-radix plans + fused pipelines vs chirp-z + unfused
# (OFDFT_MIXED_RADIX=0), beside the neighbouring powers of two.   usage: bash tools/mixed_probe.sh > gpurun_out/mixed.jsonl
run() { env $2 python bench.py --grid $1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); n=d['config']['grid'][0]
print(json.dumps({'grid': n, 'mode': sys.argv[1] or 'default', 'ms_per_eval': d['ms_per_step'], 'evals_per_s': d['value'], 'ns_per_point': round(d['ms_per_step']*1e6/n**3, 4),
                  'kernels': {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.04}}))" "$2"; }
for n in 128 120 144 160 192 256 240 250 270 288 320 384; do
  run $n ""
  case $n in 128|256) ;; *) run $n "OFDFT_MIXED_RADIX=0";; esac
done

#!/bin/bash
# rehearsal of the two slab transports with ranks SHARING one GPU (gloo for the host-side hand-shakes): library-issued peer
# copies (ipc) vs host-staged all-to-alls.  Not a scaling measurement (one GPU does all the work): it shows parity and what
# the host path costs.   usage: bash tools/ipc_probe.sh > gpurun_out/ipc_probe.txt
for t in ipc collective; do for g in 128 256; do for n in 2 4; do
  OFDFT_BENCH_SHARE_GPU=1 OFDFT_BENCH_BACKEND=gloo OFDFT_BENCH_TRANSPORT=$t timeout -k 10 200 python bench.py --gpus $n --grid $g --steps 5 --warmup 2 --no-cpu-baseline > /tmp/ipc_probe.out 2>/tmp/ipc_probe.err
  python - "$t" <<'PY'
import json, sys
lines = [l for l in open('/tmp/ipc_probe.out').read().splitlines() if l.startswith('{')]
if not lines:
    print(sys.argv[1], 'FAILED', open('/tmp/ipc_probe.err').read()[-600:])
else:
    d = json.loads(lines[-1])
    print(json.dumps({'transport': sys.argv[1], 'ranks_on_one_gpu': d['n_gpus'], 'grid': d['config']['grid'][0], 'ms_per_eval': d['ms_per_step'],
                      'parity_vs_single_gpu': d.get('parity_vs_single_gpu'), 'rel_dE_vs_reference': d['reference_check'] and d['reference_check']['rel_dE']}))
PY
done; done; done

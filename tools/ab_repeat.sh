#!/bin/bash
# one switch against the default, alternating N times on ONE box:  tools/ab_repeat.sh "ENV=VAL" [N]
run() { env $1 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-36s' % sys.argv[1], d['ms_per_step'])" "$1"; }
for r in $(seq 1 ${2:-5}); do run "A=default"; run "$1"; done

#!/bin/bash
# kernel timeline of one single-GPU closure evaluation (three streams): rocprofv3 --kernel-trace of bench.py, one evaluation as a table
#   usage: bash tools/timeline.sh <tag under gpurun_out> [bench.py args...]
TAG=${1:-tl}; shift
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/r0 -o tl -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
echo "rc=$?"
cd $REPO
python3 tools/timeline_md.py $OUT/r0 --eval 4 > $OUT/timeline.md && rm -rf $OUT/r0

#!/bin/bash
# The three rocprofv3 passes behind profiles/*_rocprof_*.md and profiles/pmc_traffic_*.json (run on the GPU box from the
# repository root): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes.  Streams serialised so that a
# kernel's begin-to-end time is its own cost.   usage: tools/profile.sh <out dir under gpurun_out> [extra bench.py arguments, e.g. --dtype f32]
set -e
OUT=$PWD/gpurun_out/${1:-prof}
shift || true
EXTRA="$@"
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OFDFT_SIDE_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $REPO/bench.py --steps 8 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o run -- python3 $REPO/bench.py --steps 8 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o run -- python3 $REPO/bench.py --steps 8 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/write.log 2>&1
cd $REPO
STAMP=$(python3 -c 'import bench; print(bench.source_stamp())')
python3 tools/rocprof_summary.py $OUT/stats $OUT/fetch $OUT/write --json $OUT/pmc_traffic.json --stamp $STAMP --evaluations 12 > $OUT/summary.md
# keep only the small artefacts (the raw traces are large)
rm -rf $OUT/stats $OUT/fetch $OUT/write

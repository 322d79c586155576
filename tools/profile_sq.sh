#!/bin/bash
# SQ-level counters of the hot kernels (run on the GPU box from the repository root): two rocprofv3 --pmc passes of the
# serialised bench command (8 SQ slots each), summarised per kernel by tools/sq_summary.py.
#   usage: tools/profile_sq.sh <out dir under gpurun_out> [extra bench.py arguments]
set -e
OUT=$PWD/gpurun_out/${1:-sq}
shift || true
EXTRA="$@"
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OFDFT_SIDE_STREAM=0
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/sqa -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/sqa.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/sqb -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/sqb.log 2>&1
cd $REPO
python3 tools/sq_summary.py $OUT/sqa $OUT/sqb > $OUT/sq_summary.md
rm -rf $OUT/sqa $OUT/sqb

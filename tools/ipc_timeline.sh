#!/bin/bash
# kernel timeline of a 2-rank slab-decomposed evaluation over the ipc transport, both ranks on the box's ONE GPU (each rank under
# its own rocprofv3 --kernel-trace, started by this shell -- never by a process that has touched the GPU).
#   usage: bash tools/ipc_timeline.sh <out dir under gpurun_out> [grid]
OUT=$PWD/gpurun_out/${1:-ipctl}
N=${2:-256}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29677 WORLD_SIZE=2 OFDFT_BENCH_SHARE_GPU=1 OFDFT_BENCH_BACKEND=gloo OFDFT_BENCH_TRANSPORT=ipc OFDFT_BENCH_NO_PARITY=1 OFDFT_BENCH_NO_SCALE512=1
RANK=0 LOCAL_RANK=0 timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/r0 -o tl -- python3 $REPO/bench.py --gpus 2 --grid $N --steps 4 --warmup 2 --no-cpu-baseline > $OUT/r0.log 2>&1 &
P0=$!
RANK=1 LOCAL_RANK=0 timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/r1 -o tl -- python3 $REPO/bench.py --gpus 2 --grid $N --steps 4 --warmup 2 --no-cpu-baseline > $OUT/r1.log 2>&1 &
P1=$!
wait $P0; echo "rank0 rc=$?"
wait $P1; echo "rank1 rc=$?"
cd $REPO
python3 tools/timeline_md.py $OUT/r0 $OUT/r1 > $OUT/timeline.md && rm -rf $OUT/r0 $OUT/r1

# final measurement artefacts of a round (run on the GPU box from the repository root, in two calls: gpurun's limit is 20 minutes):
#   bash tools/gpucall_final.sh a <tag>   serialised rocprof + PMC traffic (fp64, fp32), SQ counters
#   bash tools/gpucall_final.sh big <tag> rocprofv3 + PMC at 512^3 fp64 and 1024^3 fp32 cfg2 (install them before part c)
#   bash tools/gpucall_final.sh c <tag>   8-rank emulation probes (512^3, 256^3), x-stride probe, single-GPU timelines (fp64, fp32), 1024^3 fp32 cfg2 bench
#   bash tools/gpucall_final.sh b <tag>   bench lines (256^3 fp64 with the CPU baseline, fp32, 512^3, 64^3 cfg2), two-rank rehearsal of
#                                         `bench.py --gpus 2` (scale_512 block, both transports), two-rank ipc timeline, small-grid latencies
PART=${1:-a}
TAG=${2:-final}
mkdir -p gpurun_out
if [ "$PART" = a ]; then
  bash tools/profile.sh ${TAG}_prof > gpurun_out/${TAG}_prof.log 2>&1 || exit 1
  echo "profile f64 done"
  bash tools/profile.sh ${TAG}_prof_f32 --dtype f32 > gpurun_out/${TAG}_prof_f32.log 2>&1 || exit 1
  echo "profile f32 done"
  bash tools/profile_sq.sh ${TAG}_sq > gpurun_out/${TAG}_sq.log 2>&1 || exit 1
  bash tools/profile_sq.sh ${TAG}_sq_f32 --dtype f32 > gpurun_out/${TAG}_sq_f32.log 2>&1 || exit 1
  echo "sq done"
elif [ "$PART" = big ]; then      # rocprofv3 + PMC of the two big single-GPU workloads (before the bench lines of part c: they look the PMC files up)
  bash tools/profile.sh ${TAG}_prof_512 --grid 512 > gpurun_out/${TAG}_prof_512.log 2>&1 || exit 1
  bash tools/profile.sh ${TAG}_prof_1024_f32_cfg2 --dtype f32 --grid 1024 --cfg cfg2 > gpurun_out/${TAG}_prof_1024.log 2>&1 || exit 1
  echo "big profiles done"
elif [ "$PART" = c ]; then
  timeout -k 10 300 python tools/scale_probe.py 512 8 > gpurun_out/${TAG}_scale_probe.jsonl 2> gpurun_out/${TAG}_scale_probe_512.err || exit 1
  timeout -k 10 200 python tools/scale_probe.py 256 8 >> gpurun_out/${TAG}_scale_probe.jsonl 2> gpurun_out/${TAG}_scale_probe_256.err || exit 1
  echo "scale probes done"
  timeout -k 10 300 python tools/xpass_ab.py opts=7,1 256x256x256 256x240x256 256x250x256 512x512x512 > gpurun_out/${TAG}_x_stride_probe.jsonl 2>/dev/null
  timeout -k 10 300 python tools/xpass_ab.py f32 opts=7,5 1024x256x256 1024x240x256 >> gpurun_out/${TAG}_x_stride_probe.jsonl 2>/dev/null
  echo "stride probe rc=$?"
  bash tools/timeline.sh ${TAG}_tl64 > /dev/null 2>&1; bash tools/timeline.sh ${TAG}_tl32 --dtype f32 > /dev/null 2>&1
  echo "timelines rc=$?"
  timeout -k 10 280 python bench.py --dtype f32 --grid 1024 --cfg cfg2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_1024_f32_cfg2.json 2> gpurun_out/${TAG}_bench_1024_f32_cfg2.err
  echo "bench 1024 rc=$?"
  timeout -k 10 300 python bench.py --grid 512 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_512.json 2> gpurun_out/${TAG}_bench_512.err
  echo "bench 512 rc=$?"
else
  timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_256.json 2> gpurun_out/${TAG}_bench_256.err || exit 1
  echo "bench 256 done"
  timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_256_f32.json 2> gpurun_out/${TAG}_bench_256_f32.err || exit 1
  timeout -k 10 300 python bench.py --grid 512 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_512.json 2> gpurun_out/${TAG}_bench_512.err || exit 1
  echo "bench 512 done"
  timeout -k 10 300 python bench.py --grid 64 --cfg cfg2 --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_64_cfg2.json 2> gpurun_out/${TAG}_bench_64_cfg2.err
  echo "bench 64 rc=$?"
  OFDFT_BENCH_SHARE_GPU=1 OFDFT_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_2ranks.json 2> >(tee gpurun_out/${TAG}_bench_2ranks.err | grep --line-buffered scale_512 >&2)
  echo "two-rank rehearsal rc=$?"
  bash tools/ipc_timeline.sh ${TAG}_ipctl 256 > gpurun_out/${TAG}_ipctl.log 2>&1
  echo "ipc timeline rc=$?"
  timeout -k 10 300 python tools/latency_probe.py 16 32 53 64 > gpurun_out/${TAG}_lat.jsonl 2> gpurun_out/${TAG}_lat.err
  echo "latency rc=$?"
  timeout -k 10 300 python tools/shape_probe.py 256x256x256 240x240x240 120x120x120 270x270x270 255x255x255 129x135x127 53x53x53 > gpurun_out/${TAG}_shapes.jsonl 2> gpurun_out/${TAG}_shapes.err
  echo "shapes rc=$?"
fi

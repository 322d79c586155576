# final measurement artefacts of a round: serialised rocprof + PMC traffic (fp64, fp32), SQ counters, bench lines, small-grid probes
# usage: bash tools/gpucall_final.sh <tag>
TAG=${1:-final}
mkdir -p gpurun_out
bash tools/profile.sh ${TAG}_prof > gpurun_out/${TAG}_prof.log 2>&1 || exit 1
echo "profile f64 done"
bash tools/profile.sh ${TAG}_prof_f32 --dtype f32 > gpurun_out/${TAG}_prof_f32.log 2>&1 || exit 1
echo "profile f32 done"
bash tools/profile_sq.sh ${TAG}_sq > gpurun_out/${TAG}_sq.log 2>&1 || exit 1
echo "sq done"
cp gpurun_out/${TAG}_prof/pmc_traffic.json profiles/pmc_traffic_r02.json
cp gpurun_out/${TAG}_prof_f32/pmc_traffic.json profiles/pmc_traffic_r02_f32.json
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_256.json 2> gpurun_out/${TAG}_bench_256.err || exit 1
echo "bench 256 done"
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_256_f32.json 2> gpurun_out/${TAG}_bench_256_f32.err || exit 1
timeout -k 10 300 python bench.py --grid 512 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_512.json 2> gpurun_out/${TAG}_bench_512.err || exit 1
echo "bench 512 done"
timeout -k 10 300 python bench.py --grid 64 --cfg cfg2 --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_64_cfg2.json 2> gpurun_out/${TAG}_bench_64_cfg2.err
echo "bench 64 rc=$?"
timeout -k 10 300 python tools/latency_probe.py 16 32 64 > gpurun_out/${TAG}_lat.jsonl 2> gpurun_out/${TAG}_lat.err
echo "latency rc=$?"

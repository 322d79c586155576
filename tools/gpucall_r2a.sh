mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/r2a_counters.txt 2>&1 || true
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a_tests.log
tail -5 gpurun_out/r2a_tests.log
grep -q "rc=0" gpurun_out/r2a_tests.log || grep -q "rc=1" gpurun_out/r2a_tests.log || exit 9
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err && \
timeout -k 10 200 python bench.py --grid 512 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2a_bench512.json 2> gpurun_out/r2a_bench512.err && \
timeout -k 10 300 bash tools/profile.sh r2a_prof && \
timeout -k 10 200 bash tools/profile_sq.sh r2a_sq && \
OFDFT_BENCH_SHARE_GPU=1 OFDFT_BENCH_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --grid 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2a_gpus2.json 2> gpurun_out/r2a_gpus2.err
echo "chain rc=$?"

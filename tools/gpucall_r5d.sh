TAG=${1:-r05d}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_dist_gpu.py -x -q -m gpu > gpurun_out/${TAG}_pytest_dist.log 2>&1; echo "pytest dist rc=$?"; tail -3 gpurun_out/${TAG}_pytest_dist.log
timeout -k 10 300 python tools/scale_probe.py 512 8 > gpurun_out/${TAG}_scale_probe.jsonl 2> gpurun_out/${TAG}_scale_probe_512.err; echo "scale probe rc=$?"; cat gpurun_out/${TAG}_scale_probe.jsonl; grep "rank-0" gpurun_out/${TAG}_scale_probe_512.err | cut -c1-900

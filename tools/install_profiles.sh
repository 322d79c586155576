#!/bin/bash
# copy the small artefacts of tools/gpucall_final.sh a|b|c <tag> from gpurun_out/ into profiles/ under the round's names
# usage: bash tools/install_profiles.sh r04
T=${1:-r04}; G=gpurun_out; P=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cpif $G/${T}_prof/summary.md $P/${T}_final_rocprof_serialised.md
cpif $G/${T}_prof/pmc_traffic.json $P/pmc_traffic_${T}.json
cpif $G/${T}_prof_f32/summary.md $P/${T}_final_f32_rocprof_serialised.md
cpif $G/${T}_prof_f32/pmc_traffic.json $P/pmc_traffic_${T}_f32.json
cpif $G/${T}_sq/sq_summary.md $P/${T}_sq_counters.md
cpif $G/${T}_sq_f32/sq_summary.md $P/${T}_sq_counters_f32.md
cpif $G/${T}_bench_256.json $P/bench_${T}_final_256.json
cpif $G/${T}_bench_256_f32.json $P/bench_${T}_final_256_f32.json
cpif $G/${T}_bench_512.json $P/bench_${T}_512.json
cpif $G/${T}_bench_64_cfg2.json $P/bench_${T}_final_64_cfg2.json
cpif $G/${T}_bench_2ranks.json $P/bench_${T}_selflaunch_2gloo_ranks_256_scale512.json
cpif $G/${T}_ipctl/timeline.md $P/${T}_ipc_chunked_two_ranks_one_gpu_timeline.md
cpif $G/${T}_lat.jsonl $P/${T}_latency_small_grids.jsonl
cpif $G/${T}_shapes.jsonl $P/${T}_shapes_per_kernel.jsonl
if [ -s $G/${T}_scale_probe.jsonl ]; then cp $G/${T}_scale_probe.jsonl $P/${T}_scale_probe.jsonl; grep "rank-0" $G/${T}_scale_probe_512.err >> $P/${T}_scale_probe.jsonl; echo "  $P/${T}_scale_probe.jsonl"; fi
cpif $G/${T}_x_stride_probe.jsonl $P/${T}_x_stride_probe.jsonl
cpif $G/${T}_tl64/timeline.md $P/${T}_timeline_f64.md
cpif $G/${T}_tl32/timeline.md $P/${T}_timeline_f32.md
cpif $G/${T}_bench_1024_f32_cfg2.json $P/bench_${T}_1024_f32_cfg2.json
cpif $G/${T}_prof_512/summary.md $P/${T}_512_rocprof_serialised.md
cpif $G/${T}_prof_512/pmc_traffic.json $P/pmc_traffic_${T}_512.json
cpif $G/${T}_prof_1024_f32_cfg2/summary.md $P/${T}_1024_f32_cfg2_rocprof_serialised.md
cpif $G/${T}_prof_1024_f32_cfg2/pmc_traffic.json $P/pmc_traffic_${T}_1024_f32_cfg2.json

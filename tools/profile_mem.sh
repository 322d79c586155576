#!/bin/bash
# memory-pipeline counters of the hot kernels (GPU box, repository root): TA / TCP / TCC busy and stall counters in rocprofv3 --pmc passes
#   usage: tools/profile_mem.sh <out dir under gpurun_out> [extra bench.py arguments]
set -e
OUT=$PWD/gpurun_out/${1:-mem}
shift || true
EXTRA="$@"
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OFDFT_SIDE_STREAM=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TA_TCP_STATE_READ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys, re, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').replace('ofdft::', '')[:70]
        agg[name][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(name, r['Counter_Name'])] += 1
with open(out + '/mem_summary.md', 'w') as fh:
    names = sorted({c for v in agg.values() for c in v})
    fh.write('| kernel | ' + ' | '.join(names) + ' |\n|---|' + '---|' * len(names) + '\n')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
        fh.write('| %s | ' % k + ' | '.join('%.3g' % (v[c] / max(cnt[(k, c)], 1)) if c in v else '-' for c in names) + ' |\n')
print(open(out + '/mem_summary.md').read()[:6000])
PY
rm -rf $OUT/TA_TA_BUSY_sum $OUT/TCP_PENDING_STALL_CYCLES_sum $OUT/TCC_HIT_sum

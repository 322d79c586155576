#!/bin/bash
# rocprofv3 kernel trace + stats of tools/shape_probe.py on one shape (streams serialised by the probe's profiling passes):
#   tools/profile_shape.sh <tag> <shape> [f32]      -> gpurun_out/<tag>/stats.md (per-kernel calls, average and total time)
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $REPO/tools/shape_probe.py "$@" > $OUT/probe.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(out + '/stats.md', 'w') as fh:
    fh.write('| kernel | calls | avg us | total ms | share |\n|---|---|---|---|---|\n')
    for r in rows[:40]:
        name = re.sub(r'\(.*$', '', r['Name']).replace('void ', '').replace('ofdft::', '')
        fh.write('| %s | %s | %.1f | %.2f | %.1f%% |\n' % (name[:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
PY
grep '^{' $OUT/probe.log >> $OUT/stats.md
rm -rf $OUT/stats

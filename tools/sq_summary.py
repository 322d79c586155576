#!/usr/bin/env python3
"""Per-kernel SQ counter table from rocprofv3 --pmc passes (tools/profile_sq.sh).

usage: sq_summary.py <pass dir> [<pass dir> ...] > profiles/NAME.md

Counters are averaged over the launches of a (kernel, grid) pair.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves (MI355X_MICROARCH.md, cycle-constants table); the derived columns are ratios of such sums:
  valu%      SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES     share of a wave's life spent issuing vector ALU work
  wait%      SQ_WAIT_ANY / SQ_WAVE_CYCLES             share parked on s_waitcnt / barriers (memory, LDS returns)
  stall%     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        share stalled at issue (dependencies, busy pipes)
  ldsstall%  SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES        the LDS-issue part of the above
  conflict%  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE share of LDS-array cycles lost to bank conflicts
  occ        SQ_WAVE_CYCLES / SQ_BUSY_CYCLES x 4 ...   mean resident waves per busy SQ (of 32 per CU)
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name.replace('ofdft::', '').replace('HIP_vector_type<double, 2u>', 'cplx')


def main():
    vals = defaultdict(lambda: defaultdict(list))
    meta = {}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    key = (short(r['Kernel_Name']), int(r['Grid_Size']))
                    vals[key][r['Counter_Name']].append(float(r['Counter_Value']))
                    meta[key] = (int(r['VGPR_Count']), int(r.get('Accum_VGPR_Count') or 0), int(r['SGPR_Count']),
                                 int(r['LDS_Block_Size']), int(r['Scratch_Size']), int(r['Workgroup_Size']))
    avg = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in vals.items()}
    order = sorted(avg, key=lambda k: -avg[k].get('SQ_BUSY_CYCLES', 0.0) * len(vals[k].get('SQ_BUSY_CYCLES', [1])))
    print('| kernel | grid | launches | VGPR | LDS B/WG | scratch | waves | valu% | wait% | stall% | ldsstall% | conflict% | '
          'VALU inst/wave | LDS inst/wave | VMEM rd/wave | VMEM wr/wave | SALU inst/wave |')
    print('|' + '---|' * 17)

    def pct(a, b):
        return '%.1f' % (100.0 * a / b) if b else '-'

    for k in order:
        a = avg[k]
        if a.get('SQ_WAVES', 0) < 64 and a.get('SQ_INSTS_VALU', 0) < 1e5:
            continue
        wc = a.get('SQ_WAVE_CYCLES', 0.0)
        w = a.get('SQ_WAVES', 0.0)
        m = meta[k]
        per = (lambda c: '%.0f' % (a[c] / w) if c in a and w else '-')
        print('| %s | %d | %d | %d | %d | %d | %.0f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |' % (
            k[0][:70], k[1], len(vals[k].get('SQ_WAVES', vals[k].get('SQ_INSTS_VALU', []))), m[0] + m[1], m[3], m[4], w,
            pct(a.get('SQ_ACTIVE_INST_VALU', 0), wc), pct(a.get('SQ_WAIT_ANY', 0), wc), pct(a.get('SQ_WAIT_INST_ANY', 0), wc),
            pct(a.get('SQ_WAIT_INST_LDS', 0), wc), pct(a.get('SQ_LDS_BANK_CONFLICT', 0), a.get('SQ_LDS_IDX_ACTIVE', 0)),
            per('SQ_INSTS_VALU'), per('SQ_INSTS_LDS'), per('SQ_INSTS_VMEM_RD'), per('SQ_INSTS_VMEM_WR'), per('SQ_INSTS_SALU')))


if __name__ == '__main__':
    main()

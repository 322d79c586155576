#!/usr/bin/env python3
"""Merge the rocprofv3 --kernel-trace CSVs of the ranks of one slab-decomposed run into ONE timeline of ONE evaluation
(markdown table for profiles/), and measure how much of the exchange kernels' time ran beside compute kernels of the SAME
rank and chain (the intra-chain pipelining of the kz-chunked exchange).

usage: timeline_md.py <dir of rank 0> <dir of rank 1> [...] [--eval K] > profiles/NAME.md

An evaluation is delimited by the `sum_kernel<true>` launches of rank 0 (the first kernel of every closure evaluation).
Stream numbers are the trace's own (per process)."""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    return name.replace('ofdft::', '').replace('HIP_vector_type<double, 2u>', 'cplx')


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                st = r.get('Stream_Id') or r.get('Queue_Id') or '0'
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), st))
    rows.sort()
    return rows


COMM = ('ipc_scatter_kernel', 'ipc_stamp_kernel', 'ipc_wait_kernel', 'ipc_post_kernel', 'ipc_sum_kernel')


def main():
    which = 3
    if '--eval' in sys.argv:
        i = sys.argv.index('--eval')
        which = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    ranks = [load(d) for d in sys.argv[1:]]
    starts = [t0 for (t0, t1, nm, st) in ranks[0] if nm.startswith('sum_kernel<true>')]
    if len(starts) <= which + 1:
        sys.exit('timeline_md: fewer than %d evaluations in the trace' % (which + 2))
    lo, hi = starts[which] - 20000, starts[which + 1] - 20000
    ev = []
    for r, rows in enumerate(ranks):
        smap = {}
        for (t0, t1, nm, st) in rows:
            if lo <= t0 < hi:
                ev.append((t0, t1, r, smap.setdefault(st, len(smap)), nm))
    ev.sort()
    base = min(e[0] for e in ev)
    end = max(e[1] for e in ev)
    # overlap of a rank's scatter kernels with compute kernels of the same rank
    def overlap(a, b):
        return max(0, min(a[1], b[1]) - max(a[0], b[0]))
    if len(ranks) > 1:
        print('# Slab ranks over the ipc transport with the kz-chunked exchange, sharing ONE GPU: kernels of one evaluation, all ranks merged by time')
        print()
        print("rocprofv3 --kernel-trace of each rank (`tools/ipc_timeline.sh`), evaluation %d of the run.  The ranks share the device, so nothing here "
              "measures xGMI; the table is protocol evidence: a chain's `ipc_scatter_kernel` of chunk k (its communication stream) runs while the SAME "
              "chain's next y pass / fused x pass (its compute stream) is in flight." % which)
    else:
        print('# One single-GPU closure evaluation on its three streams: begin / end of every kernel')
        print()
        print('rocprofv3 --kernel-trace of `python3 bench.py --steps 6 --warmup 2` (`tools/timeline.sh`), evaluation %d of the run; streams NOT '
              'serialised.  The profiler inflates short kernels by a few microseconds each.' % which)
    print()
    for r in range(len(ranks)):
        sc = [e for e in ev if e[2] == r and e[4].startswith('ipc_scatter_kernel')]
        comp = [e for e in ev if e[2] == r and not e[4].startswith(COMM)]
        tot = sum(e[1] - e[0] for e in sc)
        same = 0
        for s in sc:
            # same chain = the compute stream this communication stream is fed from: take the best-overlapping compute stream
            per = {}
            for c in comp:
                o = overlap(s, c)
                if o:
                    per[c[3]] = per.get(c[3], 0) + o
            same += min(s[1] - s[0], sum(per.values())) if per else 0
        print('* rank %d: %d scatter launches, %.0f us in total, %.0f us of them (%.0f %%) concurrent with compute kernels of the same rank'
              % (r, len(sc), tot / 1e3, same / 1e3, 100.0 * same / tot if tot else 0.0))
    print('* wall of the evaluation (all ranks): %.0f us' % ((end - base) / 1e3))
    print()
    print('| begin us | end us | us | rank | stream | kernel |')
    print('|---|---|---|---|---|---|')
    for (t0, t1, r, st, nm) in ev:
        print('| %.1f | %.1f | %.1f | %d | %d | `%s` |' % ((t0 - base) / 1e3, (t1 - base) / 1e3, (t1 - t0) / 1e3, r, st, nm))


if __name__ == '__main__':
    main()

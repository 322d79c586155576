#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output into a small markdown table for profiles/.

usage: rocprof_summary.py <stats_dir> [<fetch_dir> <write_dir>] [--json OUT.json] > profiles/NAME.md

--json also writes the per-kernel-class figures (engine profiling class names, e.g. cpass_y) that bench.py reads
for `roofline.traffic`.

Per (kernel, grid size): launches, average duration from --kernel-trace, and -- when the two PMC passes are
given -- HBM traffic per launch: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), so reads = 2 * FETCH_SIZE * 1024.
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
    name = name.replace('ofdft::', '').replace('HIP_vector_type<double, 2u>', 'cplx')
    return name


def load_trace(d):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get('Grid_Size'):
                    grid = int(r['Grid_Size'])
                else:       # kernel-trace CSV: per-dimension sizes; the counter CSV has the product
                    grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y') or 1) * int(r.get('Grid_Size_Z') or 1)
                rows.append((short(r['Kernel_Name']), grid, int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    return rows


def load_counter(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] == counter:
                    acc[(short(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


# kernel symbol fragment -> the engine's profiling class (the names bench.py's `kernels` table uses)
CLASSES = [('cpass_kernel', 'cpass_y'), ('yderiv_kernel', 'yderiv'), ('ypass_xchg_kernel', 'cpass_y'),
           ('MixWgc', 'xfused_wgc'), ('MixDiv', 'xfused_div'), ('MixDerivA', 'xfused_div'), ('MixDensity', 'xfused_n'),
           ('MixScale<1>', 'xfused_lap'), ('MixScale<2>', 'xfused_lind'), ('xw_kernel', 'xfused_wgc'),
           ('zi_combine_kernel', 'zi_combine'), ('zi_wgc_kernel', 'zi_wgc'), ('zpbe2_kernel', 'zpbe'), ('zpbe_kernel', 'zpbe'),
           ('zf_powers_kernel', 'zf_powers'), ('zf_density_kernel', 'zf_density'), ('chi_grad_kernel', 'chi_grad'),
           ('sum_kernel', 'sum'), ('wgc_table_kernel', 'wgc_table'), ('reduce_partials_kernel', 'reduce'), ('reduce_rows_kernel', 'reduce'),
           ('closure_scale_kernel', 'reduce'), ('closure_scale_reduce_kernel', 'reduce'), ('axpy_kernel', 'reduce'), ('resident_closure_kernel', 'resident'),
           ('ipc_scatter_kernel', 'ipc_scatter'), ('ipc_wait_kernel', 'ipc_sync'), ('ipc_stamp_kernel', 'ipc_sync'),
           ('ipc_post_kernel', 'ipc_sync'), ('ipc_sum_kernel', 'ipc_sync'), ('ipc_abort_check_kernel', 'ipc_sync'), ('xchg_unpack_kernel', 'xchg_unpack')]
# not engine kernels: runtime copies, torch's own kernels in bench.py's copy-bandwidth probe
FOREIGN = ('__amd_rocclr', 'at::native', 'elementwise_kernel', 'vectorized_elementwise')


def klass(name):
    for key, c in CLASSES:
        if key in name:
            return c
    return None


def main():
    json_out = stamp = None
    evaluations = 12
    if '--json' in sys.argv:
        i = sys.argv.index('--json')
        json_out = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    if '--stamp' in sys.argv:          # hash of the native sources the profiled library was built from (bench.source_stamp)
        i = sys.argv.index('--stamp')
        stamp = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    if '--evaluations' in sys.argv:    # closure evaluations the profiled command ran (warm-up + timed + profiled)
        i = sys.argv.index('--evaluations')
        evaluations = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    stats = sys.argv[1]
    fetch = load_counter(sys.argv[2], 'FETCH_SIZE') if len(sys.argv) > 3 else {}
    write = load_counter(sys.argv[3], 'WRITE_SIZE') if len(sys.argv) > 3 else {}
    agg = defaultdict(list)
    for name, grid, ns in load_trace(stats):
        agg[(name, grid)].append(ns)
    tot = sum(sum(v) for v in agg.values())
    # every kernel with a share above 0.5 % must belong to a class: an unclassified one would silently drop out of the
    # per-evaluation byte sums
    lost = [(name, 100.0 * sum(v) / tot) for (name, grid), v in agg.items()
            if klass(name) is None and not any(f in name for f in FOREIGN) and sum(v) / tot > 0.005]
    if lost:
        sys.exit('rocprof_summary: unclassified kernels above 0.5 %% of the run: %r -- add them to CLASSES' % lost)
    print('| kernel | class | grid | launches | avg us | share | HBM read MB/launch (2x FETCH_SIZE) | HBM write MB/launch | traffic GB/s |')
    print('|---|---|---|---|---|---|---|---|---|')
    per_class = defaultdict(lambda: [0, 0.0, 0.0, 0.0])       # launches, ns, read MB, write MB (launch-weighted sums)
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        c = klass(name)
        if c and fetch.get((name, grid)) is not None and write.get((name, grid)) is not None:
            pc = per_class[c]
            pc[0] += len(v)
            pc[1] += sum(v)
            pc[2] += len(v) * 2 * fetch[(name, grid)] * 1024 / 1e6
            pc[3] += len(v) * write[(name, grid)] * 1024 / 1e6
        rd = fetch.get((name, grid))
        wr = write.get((name, grid))
        rd_mb = 2 * rd * 1024 / 1e6 if rd is not None else None
        wr_mb = wr * 1024 / 1e6 if wr is not None else None
        gbs = (rd_mb + wr_mb) * 1e6 / (avg * 1e-9) / 1e9 if rd_mb is not None and wr_mb is not None else None
        print('| %s | %s | %d | %d | %.1f | %.1f%% | %s | %s | %s |' % (
            name, c or '-', grid, len(v), avg / 1e3, 100.0 * sum(v) / tot,
            '%.1f' % rd_mb if rd_mb is not None else '-', '%.1f' % wr_mb if wr_mb is not None else '-',
            '%.0f' % gbs if gbs is not None else '-'))
    if per_class:
        total_mb = sum(v[2] + v[3] for v in per_class.values())
        print()
        print('HBM traffic of all engine kernels: %.2f GB per evaluation (%d evaluations in the run); source stamp %s'
              % (total_mb / evaluations / 1e3, evaluations, stamp))

    if json_out:
        import json
        out = {'source': 'rocprofv3 --kernel-trace (durations) and two --pmc passes (FETCH_SIZE, WRITE_SIZE) of '
                         '`OFDFT_SIDE_STREAM=0 python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline`; '
                         'reads = 2 x FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md HBM section), '
                         'writes = WRITE_SIZE KiB; per launch',
               'source_stamp': stamp, 'evaluations': evaluations,
               'kernels': {c: {'launches': v[0], 'avg_us': round(v[1] / v[0] / 1e3, 2), 'read_MB': round(v[2] / v[0], 1),
                               'write_MB': round(v[3] / v[0], 1)} for c, v in per_class.items()}}
        with open(json_out, 'w') as fh:
            json.dump(out, fh, indent=1)


if __name__ == '__main__':
    main()

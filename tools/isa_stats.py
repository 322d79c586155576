#!/usr/bin/env python3
"""Static instruction mix of gfx950 kernels from a hipcc -S listing (build-time aid; no GPU needed).

usage: isa_stats.py engine.s <substring of the demangled-ish kernel symbol> [...]
Emit the listing with:
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -I include --offload-device-only -S -o engine.s professad_amd/csrc/engine.hip
"""
import collections
import re
import subprocess
import sys


def kernels(path):
    """symbol -> (list of instruction mnemonics, dict of metadata)"""
    out = {}
    cur, body = None, []
    meta = {}
    with open(path) as fh:
        for line in fh:
            m = re.match(r'^(_Z\w+):\s*(;.*)?$', line)
            if m and cur is None:
                cur, body, meta = m.group(1), [], {}
                continue
            if cur is None:
                continue
            if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
                pass
            s = line.strip()
            if s.startswith('s_endpgm'):
                body.append('s_endpgm')
            m2 = re.match(r'^\t([sv]_\w+|ds_\w+|buffer_\w+|global_\w+|flat_\w+|scratch_\w+)\b', line)
            if m2:
                body.append(m2.group(1))
            m3 = re.match(r'^; (NumVgprs|NumAgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte): (\d+)', line) or \
                re.match(r'^; (\w+): (\d+)', line)
            if m3:
                meta[m3.group(1)] = int(m3.group(2))
            if line.startswith('\t.section') or line.startswith('\t.text'):
                if body:
                    out[cur] = (body, meta)
                cur = None
    return out


def demangle(names):
    p = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def classify(op):
    if op.startswith('v_') and ('_f64' in op or op in ('v_fma_f64',)):
        if any(t in op for t in ('rcp', 'rsq', 'sqrt', 'div_', 'frexp', 'ldexp', 'trig', 'fract', 'rndne', 'floor', 'ceil')):
            return 'v_f64_special'
        return 'v_f64_arith'
    if op.startswith('v_') and '_f32' in op:
        return 'v_f32'
    if op.startswith('v_'):
        return 'v_other'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 's_waitcnt'
    if op.startswith('s_barrier'):
        return 's_barrier'
    return 'salu'


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    ks = kernels(path)
    dm = demangle(list(ks))
    for sym, (body, meta) in ks.items():
        name = dm.get(sym, sym)
        if pats and not any(p in name for p in pats):
            continue
        c = collections.Counter(classify(op) for op in body)
        top = collections.Counter(body).most_common(12)
        print('%s\n   %s' % (name[:160], {k: meta[k] for k in ('NumVgprs', 'NumAgprs', 'NumSgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize') if k in meta}))
        print('   total %d  ' % len(body) + '  '.join('%s %d' % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])))
        print('   top: ' + ', '.join('%s %d' % kv for kv in top))


if __name__ == '__main__':
    main()

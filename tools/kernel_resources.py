#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of every translation unit of csrc/ for gfx950, from
`hipcc -Rpass-analysis=kernel-resource-usage` (device-only compiles in parallel, no GPU needed).

    python tools/kernel_resources.py [--scratch-only] [--filter k_rollout]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'blockpuzzle-gym_amd', 'csrc')
SOURCES = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
FIELDS = (('sgpr', r'SGPRs'), ('vgpr', r'VGPRs'), ('agpr', r'AGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
          ('occupancy', r'Occupancy \[waves/SIMD\]'), ('sgpr_spill', r'SGPRs Spill'), ('vgpr_spill', r'VGPRs Spill'),
          ('lds', r'LDS Size \[bytes/block\]'))


def demangle(names):
    import shutil
    filt = shutil.which('c++filt') or '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'
    if not os.path.exists(filt):
        return names
    out = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    return out if len(out) == len(names) else names


def _remarks(src, hipcc, extra_flags=()):
    with tempfile.TemporaryDirectory() as d:
        cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-c', '-fPIC', '-ffp-contract=off', *extra_flags,
               '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-o', os.path.join(d, 'dev.o'), src]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-4000:])
    return r.stderr


def resources(sources=None, hipcc='/opt/rocm/bin/hipcc', extra_flags=()):
    """[{name, unit, sgpr, vgpr, agpr, scratch, occupancy, sgpr_spill, vgpr_spill, lds}] for every kernel."""
    from concurrent.futures import ThreadPoolExecutor
    sources = list(sources or SOURCES)
    with ThreadPoolExecutor(min(len(sources), os.cpu_count() or 4)) as ex:
        texts = list(ex.map(lambda s: _remarks(s, hipcc, extra_flags), sources))
    blocks = []
    for src, text in zip(sources, texts):
        blocks += [(os.path.basename(src), b) for b in re.split(r'remark: [^\n]*Function Name: ', text)[1:]]
    rows = []
    for unit, b in blocks:
        row = {'mangled': b.split()[0], 'unit': unit}
        for key, pat in FIELDS:
            m = re.search(r'remark: [^\n]*?' + pat + r': (\d+)', b)
            row[key] = int(m.group(1)) if m else None
        rows.append(row)
    for row, nm in zip(rows, demangle([r_['mangled'] for r_ in rows])):
        row['name'] = re.sub(r'\(.*', '', nm)
    return rows


def main():
    rows = resources()
    flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
    for r in rows:
        if flt and flt not in r['name']:
            continue
        if '--scratch-only' in sys.argv and not r['scratch']:
            continue
        print('%-110s vgpr %3s sgpr %3s scratch %4s (spill v %s s %s) occ %s lds %s' % (
            r['name'][:110], r['vgpr'], r['sgpr'], r['scratch'], r['vgpr_spill'], r['sgpr_spill'], r['occupancy'], r['lds']))
    print('%d kernels, %d with scratch' % (len(rows), sum(1 for r in rows if r['scratch'])))


if __name__ == '__main__':
    main()

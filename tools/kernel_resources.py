#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of every translation unit of csrc/ for gfx950, from
`hipcc -Rpass-analysis=kernel-resource-usage` (device-only compiles in parallel, no GPU needed).

    python tools/kernel_resources.py [--scratch-only] [--filter k_rollout]
    python tools/kernel_resources.py --spill-sites [unit ...]      where the SGPR spills of a unit's kernels sit (loop depth)
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


def spill_sites(src, hipcc='/opt/rocm/bin/hipcc'):
    """Where the SGPR spills of every kernel of one translation unit are: {kernel: {'writes': {depth: n}, 'reads': {depth: n}}},
    depth = loop nesting of the basic block (0 = outside every loop), from the device assembly.  An SGPR spill is a
    v_writelane_b32 into a constant lane of a VGPR and its restore a v_readlane_b32 from one; at depth 0 -- kernel arguments the
    epilogue needs, parked in a VGPR lane while the loops run -- they cost a handful of instructions per wave per launch; inside
    a loop they would cost issue slots on every trip."""
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, 'dev.s')
        cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '--cuda-device-only', '-S',
               '-o', asm, src]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-4000:])
        text = open(asm).read()
    out, names = {}, []
    for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)^\.Lfunc_end', text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        depth, sites = 0, {'writes': {}, 'reads': {}}
        lines = body.split('\n')
        for i, ln in enumerate(lines):
            if re.match(r'\.LBB\d+_\d+:', ln):
                # the block's loop annotation is on the label line and, for nested loops, the comment lines right behind it
                note = ln + ' '.join(lines[i + 1:i + 4] if ';' in ln else [])
                dm = re.findall(r'Depth=(\d+)', ' '.join([ln] + [x for x in lines[i + 1:i + 4] if x.lstrip().startswith(';')]))
                depth = max(int(x) for x in dm) if dm else 0
                continue
            t = ln.strip()
            if re.match(r'v_writelane_b32 v\d+, s\d+, \d+', t):
                sites['writes'][depth] = sites['writes'].get(depth, 0) + 1
            elif re.match(r'v_readlane_b32 s\d+, v\d+, \d+', t):
                sites['reads'][depth] = sites['reads'].get(depth, 0) + 1
        if sites['writes'] or sites['reads']:
            out[name] = sites
            names.append(name)
    for name, nm in zip(names, demangle(names)):
        out[name]['name'] = re.sub(r'\(.*', '', nm)
    return out


def main():
    if '--spill-sites' in sys.argv:
        units = [a for a in sys.argv[sys.argv.index('--spill-sites') + 1:] if not a.startswith('-')] or ['rollout']
        for u in units:
            sites = spill_sites(os.path.join(CSRC, u + '.hip'))
            inside = 0
            for k, v in sorted(sites.items(), key=lambda kv: kv[1]['name']):
                deep = sum(n for d, n in list(v['writes'].items()) + list(v['reads'].items()) if d > 0)
                inside += deep
                print('%-100s spill writes by loop depth %s  restores %s' % (v['name'][:100], v['writes'], v['reads']))
            print('%s: %d kernels with SGPR spills, %d spill / restore instructions inside a loop' % (u, len(sites), inside))
        return
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

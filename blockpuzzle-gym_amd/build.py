"""In-tree build of the HIP extension (hipcc, gfx950 only): every translation unit of csrc/ is compiled on its
own, in parallel, and the objects are linked into lib/libbpgym.so."""
import os
import shutil
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# the long compiles first (8 units run at once on the build container's 8 cores)
UNITS = ('returns_stream', 'returns_stream_rt', 'rollout', 'returns', 'episode', 'step', 'reward', 'obs_reset_place', 'abi', 'comm')
SOURCES = [os.path.join(CSRC, u + '.hip') for u in UNITS]
HEADERS = [os.path.join(CSRC, 'state.hip.h'), os.path.join(CSRC, 'hooks.hip.h'), os.path.join(CSRC, 'returns.hip.h'), os.path.join(CSRC, 'returns_lane.hip.h'),
           os.path.join(os.path.dirname(_HERE), 'include', 'bpgym.h')]
OBJ_DIR = os.path.join(_HERE, 'lib', 'obj')
OUT = os.path.join(_HERE, 'lib', 'libbpgym.so')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-Wall']

def hipcc():
    return shutil.which('hipcc') or '/opt/rocm/bin/hipcc'

def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)

def needs_build():
    return _stale(OUT, SOURCES + HEADERS)

def build(force=False, verbose=False, jobs=None, extra_flags=(), out=None):
    """Compile what is stale (everything with force=True) and link.  extra_flags / out: a variant of the library
    for an A/B measurement (tools/), built beside the product library from the same sources."""
    out = out or OUT
    variant = out != OUT
    obj_dir = OBJ_DIR if not variant else out + '.obj'
    if not force and not variant and not needs_build():
        return out
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + '.o')
        if not force and not _stale(obj, [src] + HEADERS):
            return src, 0.0
        cmd = [cc] + FLAGS + list(extra_flags) + ['-c', '-o', obj, src]
        if verbose:
            print(' '.join(cmd), flush=True)
        t0 = time.time()
        subprocess.check_call(cmd)
        return src, time.time() - t0

    t0 = time.time()
    with ThreadPoolExecutor(jobs or min(len(SOURCES), os.cpu_count() or 4)) as ex:
        times = list(ex.map(compile_one, SOURCES))
    objs = [os.path.join(obj_dir, os.path.basename(s)[:-4] + '.o') for s in SOURCES]
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    if verbose:
        print('built %s in %.1f s (%s)' % (os.path.relpath(out, os.path.dirname(_HERE)), time.time() - t0,
                                           ', '.join('%s %.0f s' % (os.path.basename(s)[:-4], dt) for s, dt in times)))
    return out

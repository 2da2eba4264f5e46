"""In-tree build of the HIP extension (hipcc, gfx950 only)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, 'csrc', 'bpgym.hip')
INC = os.path.join(os.path.dirname(_HERE), 'include', 'bpgym.h')
OUT = os.path.join(_HERE, 'lib', 'libbpgym.so')


def hipcc():
    return shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in (SRC, INC))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-shared', '-fPIC',
           '-ffp-contract=off', '-Wall', '-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT

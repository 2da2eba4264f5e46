import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The built libraries are git-ignored: build them in-tree when a fresh checkout has
    none (hipcc cross-compiles gfx950 without a GPU; the oracle needs only gcc)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bpg_build', os.path.join(ROOT, 'blockpuzzle-gym_amd', 'build.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    if b.needs_build():
        b.build()
    from oracle import bpg_oracle
    bpg_oracle.lib()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN

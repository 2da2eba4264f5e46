"""CPU-only checks of the C-ABI library and the host logic: the library loads,
exports every symbol include/bpgym.h declares, and its host-side tables match
the reference-derived fixtures.  No kernel is launched here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import blockpuzzle_gym_amd as bpg
from blockpuzzle_gym_amd import _lib
from util import GOLDEN, VARIANTS, ENV_IDS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'bpgym.h')).read()
    declared = sorted(set(re.findall(r'\b(bpg_[a-z0-9_]+)\s*\(', header)))
    assert len(declared) >= 25
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_lib.SIGNATURES) == declared, 'ctypes binding and header disagree'


def test_registry_mirrors_reference():
    reg = json.load(open(os.path.join(GOLDEN, 'registry.json')))
    assert list(bpg.registry) == [r['id'] for r in reg]
    for r in reg:
        spec = bpg.registry[r['id']]
        assert spec.max_episode_steps == r['max_episode_steps'] == 50
        assert spec.kwargs == r['kwargs']
        assert spec.entry_point == r['entry_point']
    L = _lib.lib()
    assert L.bpg_variant_from_id(b'haha-v0') == -1        # README.md:21 id is never registered
    with pytest.raises(KeyError):
        bpg.make('haha-v0')


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_variant_tables_match_reference(env_id):
    v = VARIANTS[env_id]
    L = _lib.lib()
    info = _lib.variant_info(L.bpg_variant_from_id(env_id.encode()))
    assert info.env_id.decode() == env_id
    assert (info.num_objs, info.dimo, info.dimg, info.dimu) == (v['num_objs'], v['dimo'], v['dimg'], 4)
    assert bool(info.block_gripper) == v['block_gripper']
    assert list(info.obj_colors[:info.num_objs]) == v['obj_colors']
    assert list(info.goal[:info.dimg]) == v['goal']
    assert bool(info.desired_goal_is_float) == (v['desired_goal_dtype'] == 'float64')
    assert bool(info.set_test_supported) == (v['set_test'] == 'ok')
    status = {'ok': _lib.OK, 'NotImplementedError': _lib.ERR_UNSUPPORTED, 'AttributeError': _lib.ERR_ATTRIBUTE}
    assert info.increase_difficulty_status == status[v['increase_difficulty']]
    # packed goal planes agree with the goal matrix
    side = info.side
    for i in range(side):
        for j in range(side):
            bit = 1 << L.bpg_pair_index(i, j)
            g = v['goal'][i * side + j]
            assert bool(info.want_touch & bit) == (g == 1)
            assert bool(info.want_never & bit) == (g == -1)
            assert info.valid_pairs & bit


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_geom2objid_over_real_geom_names(env_id):
    names = bpg.default_geom_names(env_id)
    for k, m in enumerate(VARIANTS[env_id]['models']):
        assert names[k] == m['geom_names']
        got = [bpg.geom2objid(n) for n in names[k]]
        assert [-1 if g is None else g for g in got] == m['id2obj']


def test_pair_index_is_a_bijection():
    L = _lib.lib()
    seen = {}
    for i in range(6):
        for j in range(6):
            p = L.bpg_pair_index(i, j)
            assert 0 <= p < 21 and p == L.bpg_pair_index(j, i)
            seen.setdefault(p, (min(i, j), max(i, j)))
            assert seen[p] == (min(i, j), max(i, j))
    assert len(seen) == 21
    assert L.bpg_pair_index(6, 0) == -1


def test_error_reporting_without_gpu_calls():
    L = _lib.lib()
    info = _lib.VariantInfo()
    assert L.bpg_get_variant_info(99, C.byref(info)) == _lib.ERR_ARG
    assert b'variant' in L.bpg_last_error()
    h = C.c_void_p()
    assert L.bpg_create(C.byref(h), 99, 4, 0, 50, 0, 0) == _lib.ERR_ARG
    assert L.bpg_create(C.byref(h), 0, 0, 0, 50, 0, 0) == _lib.ERR_ARG
    assert L.bpg_create(C.byref(h), 0, 4, 0, 5000, 0, 0) == _lib.ERR_ARG


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'blockpuzzle-gym_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f
                assert 'bpg_oracle' not in text, f


def test_shard_ranges_cover_the_batch():
    from blockpuzzle_gym_amd.distributed import shard_range
    for total, world in [(4194304, 8), (1048576, 4), (10, 3), (7, 8), (1, 1)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_row_stride_helper_is_a_pure_host_function():
    """bpg_row_stride (layout of the time-major streams) needs no GPU: contiguous rows
    below 2^20 envs, above that an odd multiple of 256 elements (rows shifted by 1 KB against each other)."""
    L = _lib.lib()
    assert L.bpg_row_stride(0) == 0 and L.bpg_row_stride(-5) == 0
    for n in (1, 3, 65536, (1 << 20) - 1):
        assert L.bpg_row_stride(n) == n
    assert L.bpg_row_stride(1 << 20) == (1 << 20) + 256
    assert L.bpg_row_stride((1 << 20) + 256) == (1 << 20) + 256          # already an odd multiple
    for n in (1 << 20, (1 << 20) + 1, 3 << 19, 1 << 22, (1 << 22) + 2, 1 << 25, 45_000_002):
        ld = L.bpg_row_stride(n)
        assert n <= ld < n + 512 and ld % 256 == 0 and (ld // 256) % 2 == 1
    # the struct the header declares and the ctypes mirror agree on the new trailing fields
    a = _lib.RolloutArgs(T=1, ld_masks=7, ld_out=9)
    assert (a.ld_masks, a.ld_out) == (7, 9) and C.sizeof(_lib.RolloutArgs) == 16 + 9 * 8 + 16 + 16 + 8 + 8 + 8
    # BPG_ROLLOUT_ARGS_INIT: the mirror fills in BPG_ARGS_TAG(its own size); the minimum the library accepts ends behind gamma
    assert a.struct_size == _lib.args_tag(C.sizeof(_lib.RolloutArgs)) == 0xB5A90000 | 144
    assert _lib.RolloutArgs.gamma.offset + 8 == _lib.ROLLOUT_ARGS_MIN_SIZE == 120
    # the round-3 names survive as stubs: a stale caller gets an error code whatever it passes
    assert L.bpg_rollout() == _lib.ERR_UNSUPPORTED and b'bpg_rollout_v2' in L.bpg_last_error()
    assert L.bpg_step_record() == _lib.ERR_UNSUPPORTED and b'bpg_step_record_v2' in L.bpg_last_error()


def test_every_kernel_stub_has_a_device_kernel_of_the_same_name():
    """hipcc mangles a kernel's name twice (host stub, device symbol).  A type without linkage in the signature
    (an unnamed enum in a template argument, or in a template-DEPENDENT parameter type such as
    std::conditional_t<SRC == SRC_SYNTH, ...>) is numbered differently by the two passes -- `$_0` on the host, `$_1` on the
    device -- and the launch dies at run time with "Cannot find Symbol", only on the GPU box (it did, in round 5).  Check the
    built library here: the name the host registers must be the name of a kernel DESCRIPTOR (`<name>.kd`, which only the
    device code object has -- the bare name also occurs as the host's registration string)."""
    data = open(_lib.LIB_PATH, 'rb').read()
    stubs = set(re.findall(rb'_ZN4bpgk\d+__device_stub__([A-Za-z0-9_$.]+)', data))
    assert len(stubs) > 100
    assert not [s for s in stubs if b'$_' in s][:3], 'a type without linkage in a kernel signature'
    missing = []
    for name in stubs:
        # the device symbol is _ZN4bpgk<len><name...>: same text after the length-prefixed function name
        m = re.match(rb'(k_[a-z0-9_]+?)(I.*|E.*)$', name)
        base, rest = m.group(1), m.group(2)
        dev = b'_ZN4bpgk' + str(len(base)).encode() + base + rest
        if dev + b'.kd' not in data:
            missing.append(name[:120])
    assert not missing, missing[:3]


def test_no_kernel_uses_scratch_memory():
    """Register spills to scratch are silent performance bugs (round 2 shipped seven k_rollout instantiations
    with 36-328 bytes per lane because their launch bound promised 1024-lane blocks they were never launched
    with).  Compile the device code with -Rpass-analysis=kernel-resource-usage and require ScratchSize == 0."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('kernel_resources', os.path.join(ROOT, 'tools', 'kernel_resources.py'))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = kr.resources()
    assert len(rows) > 100
    bad = [(r['name'], r['scratch']) for r in rows if r['scratch']]
    assert not bad, bad[:5]
    # the fused rollout is never launched with more lanes than its launch bound, and holds its registers
    # (template arguments: ..., LB = the block size the instantiation is launched with, OKB, MULTI, RET): 1024 lanes are
    # 4 waves per SIMD = 128 registers each; the 256-lane shapes must keep at least two waves per SIMD
    seen = 0
    for r in rows:
        m = re.search(r'k_rollout<(.*)>', r['name'])
        if m:
            lb = int(m.group(1).split(',')[6])
            assert lb in (256, 1024) and r['vgpr'] <= (128 if lb == 1024 else 256), r
            seen += 1
    assert seen >= 100


def test_sgpr_spills_of_the_headline_kernel_stay_outside_its_loops():
    """The fused rollout keeps ~100 scalars alive (row addresses of the unrolled steps, goal planes, strides) and the compiler
    parks the kernel arguments that only the epilogue needs (the success-count words) in lanes of a VGPR: "SGPRs Spill: 10" in
    the resource table.  That is free as long as the spill writes sit in the prologue and the restores behind the loops; inside
    the step loop they would cost issue slots on every trip.  tools/kernel_resources.py --spill-sites reads the loop depth of
    every v_writelane / v_readlane out of the device assembly: the headline instantiation (masks, 2 envs per lane, reward +
    flags, reset, 1024 lanes) must have none at depth > 0, and no instantiation more than a handful in its innermost loop."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('kernel_resources', os.path.join(ROOT, 'tools', 'kernel_resources.py'))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    sites = kr.spill_sites(os.path.join(ROOT, 'blockpuzzle-gym_amd', 'csrc', 'rollout.hip'))
    by_name = {v['name']: v for v in sites.values()}
    head = [v for k, v in by_name.items() if 'k_rollout<0, 2, 3, 1, 5, false, 1024, 0>' in k]
    assert len(head) == 1, sorted(by_name)[:5]
    assert set(head[0]['writes']) <= {0} and set(head[0]['reads']) <= {0}, head[0]
    for name, v in by_name.items():
        # MODE_MULTI (4th argument 2) wraps one more loop around everything: its step loop is depth 3, the others' depth 2
        multi = re.search(r'k_rollout<\d+, \d+, \d+, 2,', name) is not None
        inner = 3 if multi else 2
        n_inner = sum(n for d, n in list(v['writes'].items()) + list(v['reads'].items()) if d >= inner)
        assert n_inner <= 4, (name, v)


def test_register_with_gym_against_a_recording_stand_in(monkeypatch):
    """gym is not installable here; a stand-in `gym.envs.registration.register` records what register_with_gym hands
    it: the seven ids of gym_blocks/__init__.py:6-53, max_episode_steps = 50, kwargs {'reward_type': 'sparse'},
    callable entry points.  Without gym the function registers nothing."""
    import sys
    import types
    from blockpuzzle_gym_amd import registration as R
    for name in ('gym', 'gym.envs', 'gym.envs.registration'):
        monkeypatch.delitem(sys.modules, name, raising=False)
    assert R.register_with_gym() == []
    calls = []
    reg = types.ModuleType('gym.envs.registration')
    reg.register = lambda **kw: calls.append(kw)
    gym, envs = types.ModuleType('gym'), types.ModuleType('gym.envs')
    gym.envs, envs.registration = envs, reg
    for name, mod in (('gym', gym), ('gym.envs', envs), ('gym.envs.registration', reg)):
        monkeypatch.setitem(sys.modules, name, mod)
    ids = R.register_with_gym()
    want = json.load(open(os.path.join(GOLDEN, 'registry.json')))
    assert ids == [r['id'] for r in want] == [c['id'] for c in calls]
    for c, r in zip(calls, want):
        assert c['max_episode_steps'] == r['max_episode_steps'] == 50 and c['kwargs'] == r['kwargs']
        assert callable(c['entry_point'])
    assert R.register_with_gym(prefix='Amd') [0] == 'AmdGripperTouch-v0'


def test_header_is_plain_c_and_its_structs_have_the_layout_ctypes_assumes(tmp_path):
    """The boundary is a C ABI: include/bpgym.h must compile as C99 on its own (no C++, no HIP, no torch types), and
    the structs a C compiler lays out must be what blockpuzzle-gym_amd/_lib.py mirrors, field by field."""
    import subprocess
    pairs = [('bpg_rollout_args', _lib.RolloutArgs), ('bpg_step_args', _lib.StepArgs), ('bpg_variant_info', _lib.VariantInfo),
             ('bpg_phys_state', _lib.PhysState), ('bpg_placement_args', _lib.PlacementArgs), ('bpg_curriculum', _lib.Curriculum)]
    # the constant initialisers (C statics) carry the same tag as the function-style ones
    body = ('  static bpg_rollout_args sr = BPG_ROLLOUT_ARGS_INIT_STATIC; static bpg_step_args ss = BPG_STEP_ARGS_INIT_STATIC;\n'
            '  bpg_rollout_args ar = BPG_ROLLOUT_ARGS_INIT; bpg_step_args as = BPG_STEP_ARGS_INIT;\n'
            '  if (sr.struct_size != ar.struct_size || ss.struct_size != as.struct_size || sr.d_episode_fail_count || ss.d_return_acc) return 1;\n'
            '  printf("tag.rollout %u\\n", sr.struct_size); printf("tag.step %u\\n", ss.struct_size);\n')
    for cname, ct in pairs:
        body += '  printf("%s.size %%zu\\n", sizeof(%s));\n' % (cname, cname)
        for f in ct._fields_:
            body += '  printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (cname, f[0], cname, f[0])
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bpgym.h"\nint main(void) {\n' + body + '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src),
                    '-o', str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(out['tag.rollout']) == _lib.args_tag(C.sizeof(_lib.RolloutArgs)) and int(out['tag.step']) == _lib.args_tag(C.sizeof(_lib.StepArgs))
    for cname, ct in pairs:
        assert int(out[cname + '.size']) == C.sizeof(ct), cname
        for f in ct._fields_:
            assert int(out['%s.%s' % (cname, f[0])]) == getattr(ct, f[0]).offset, (cname, f[0])


def test_every_entry_point_is_exercised_somewhere():
    """No entry point of the header may exist only as a declaration: each one is either called by the host package
    (whose paths the GPU tests drive) or named by a test directly."""
    import glob
    header = open(os.path.join(ROOT, 'include', 'bpgym.h')).read()
    declared = sorted(set(re.findall(r'\b(bpg_[a-z0-9_]+)\s*\(', header)))
    tests = ''.join(open(f).read() for f in glob.glob(os.path.join(ROOT, 'tests', '*.py')) if not f.endswith('test_abi_cpu.py'))
    pkg = ''.join(open(f).read() for f in glob.glob(os.path.join(ROOT, 'blockpuzzle-gym_amd', '*.py')))
    idle = [n for n in declared
            if not re.search(r'\b%s\b' % n, tests) and len(re.findall(r'\b%s\b' % n, pkg)) < 2]      # 1 = the signature table only
    assert idle == ['bpg_version'], idle            # bpg_version: below
    assert C.CDLL(_lib.LIB_PATH).bpg_version() >= 105


def test_lazy_success_history_reads_like_a_deque_of_floats():
    """The async_stats worker's success_history (rollout.py:37, :174-184 read it with np.mean): entries resolve on first
    read, from the difference of two running totals; host logic only."""
    from blockpuzzle_gym_amd.rollout import _History, _LazyRate

    class Snap:
        reads = 0

        def __init__(self, v):
            self.v = v

        def value(self):
            Snap.reads += 1
            return self.v
    h = _History(maxlen=3)
    totals = [Snap(10), Snap(25), Snap(25), Snap(65)]
    h.append(_LazyRate(0, totals[0], 100))
    for a, b in zip(totals, totals[1:]):
        h.append(_LazyRate(a, b, 100))
    assert len(h) == 3 and Snap.reads == 0              # nothing read yet; the first entry fell out of the window
    assert np.mean(h) == pytest.approx((0.15 + 0.0 + 0.40) / 3)
    assert list(h) == [0.15, 0.0, 0.40] and h[-1] == 0.40 and h == [0.15, 0.0, 0.40]
    assert all(type(x) is float for x in h)
    h.append(0.5)
    assert list(h) == [0.0, 0.40, 0.5]
    h.clear()
    assert len(h) == 0

"""Pins the CPU oracle (oracle/bpg_oracle.c) against fixtures recorded from the
reference's own code (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import bpg_oracle as O

G = os.path.join(os.path.dirname(__file__), 'golden')
VARIANTS = json.load(open(os.path.join(G, 'variants.json')))
ENV_IDS = list(VARIANTS)


def short(env_id):
    return env_id.split('-')[0]


def test_registry_matches_reference():
    # gym_blocks/__init__.py:6-53
    reg = json.load(open(os.path.join(G, 'registry.json')))
    assert [r['id'] for r in reg] == ENV_IDS
    assert all(r['max_episode_steps'] == 50 for r in reg)
    assert all(r['kwargs'] == {'reward_type': 'sparse'} for r in reg)
    assert set(O.VARIANTS) == set(ENV_IDS)


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_geom2objid_and_goal(env_id):
    v = VARIANTS[env_id]
    L = O.lib()
    for m in v['models']:
        got = [L.orc_geom2objid(None if n is None else n.encode()) for n in m['geom_names']]
        assert got == m['id2obj']
    e = O.OracleEnv(env_id)
    assert e.e.num_objs == v['num_objs']
    assert list(e.e.obj_colors[:v['num_objs']]) == v['obj_colors']
    e.reset()
    assert e.dimg == v['dimg']
    assert e.goal().tolist() == v['goal']


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_step_traces(env_id):
    """Full reference env.step()/reset() sequences, incl. Q1 persistence across
    resets, -0.0 rewards, the is_success latch and TimeLimit done."""
    v = VARIANTS[env_id]
    d = np.load(os.path.join(G, 'trace_%s.npz' % short(env_id)))
    e = O.OracleEnv(env_id, v['max_episode_steps'])
    id2objs = [m['id2obj'] for m in v['models']]
    neg0 = 0
    for ep, L in enumerate(d['lens']):
        ag0 = e.reset(num_grey=int(d['num_objs'][ep]) - 4)
        assert e.e.num_objs == d['num_objs'][ep]
        np.testing.assert_array_equal(ag0, d['ag_reset'][ep])
        np.testing.assert_array_equal(e.goal(), d['g'][ep])
        id2obj = id2objs[int(d['model_idx'][ep])]
        for t in range(L):
            con = d['contacts'][ep, t, :d['ncon'][ep, t]]
            ag, r, done, succ = e.step(id2obj, con)
            np.testing.assert_array_equal(ag, d['ag'][ep, t])
            assert np.float32(r).view(np.uint32) == d['reward_bits'][ep, t]
            assert done == bool(d['done'][ep, t])
            assert succ == bool(d['success'][ep, t])
            neg0 += int(d['reward_bits'][ep, t] == 0x80000000)
            # mask form of the same step gives the same matrix
            assert O.contacts_to_mask(id2obj, con) < (1 << 21)
    assert neg0 > 0, 'fixture must exercise the -0.0 success reward'


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_step_traces_mask_form(env_id):
    v = VARIANTS[env_id]
    d = np.load(os.path.join(G, 'trace_%s.npz' % short(env_id)))
    e = O.OracleEnv(env_id, v['max_episode_steps'])
    id2objs = [m['id2obj'] for m in v['models']]
    for ep, L in enumerate(d['lens']):
        e.reset(num_grey=int(d['num_objs'][ep]) - 4)
        id2obj = id2objs[int(d['model_idx'][ep])]
        for t in range(L):
            m = O.contacts_to_mask(id2obj, d['contacts'][ep, t, :d['ncon'][ep, t]])
            ag, r, done, succ = e.step_mask(m)
            np.testing.assert_array_equal(ag, d['ag'][ep, t])
            assert np.float32(r).view(np.uint32) == d['reward_bits'][ep, t]
            assert (done, succ) == (bool(d['done'][ep, t]), bool(d['success'][ep, t]))


def test_compute_reward_batched():
    d = np.load(os.path.join(G, 'compute_reward.npz'))
    keys = sorted({k.rsplit('_', 1)[0] for k in d.files})
    assert len(keys) == 5
    for k in keys:
        ag, g, g2 = d[k + '_ag'], d[k + '_g'], d[k + '_g2']
        np.testing.assert_array_equal(O.compute_reward(ag, g).view(np.uint32), d[k + '_rbits'])
        np.testing.assert_array_equal(O.compute_reward(ag, g2).view(np.uint32), d[k + '_rbits2'])
        n2 = ag.shape[1]
        r3 = O.compute_reward(ag[:240].reshape(4, 60, n2), g2[:240].reshape(4, 60, n2))
        np.testing.assert_array_equal(r3.view(np.uint32), d[k + '_rbits3'])
        assert (d[k + '_rbits'] == 0x80000000).any() and (d[k + '_rbits'] == 0xbf800000).any()


def test_out_of_table_and_constants():
    m = json.load(open(os.path.join(G, 'misc.json')))
    L = O.lib()
    pts, exp = m['out_of_table']['points'], m['out_of_table']['expected']
    assert [bool(L.orc_out_of_table(x, y)) for x, y in pts] == exp
    assert any(exp) and not all(exp)
    c = m['constants']
    assert (c['GREY'], c['RED'], c['GREEN'], c['BLUE']) == (0, 1, 2, 3)


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_rollout_student_episodes(env_id):
    """Reference RolloutStudent.generate_rollouts (rollout.py:75-172) episode
    dicts over consecutive rollouts -- ag/g/info_is_success/success_rate."""
    v = VARIANTS[env_id]
    d = np.load(os.path.join(G, 'rollout_%s.npz' % short(env_id)))
    B, T = int(d['B']), int(d['T'])
    envs = None
    for r in range(len(d['modes'])):
        id2objs = [m['id2obj'] for m in v['models']]
        masks = np.zeros((T, B), np.uint32)
        for b in range(B):
            id2obj = id2objs[int(d['r%d_model_idx' % r][b])]
            for t in range(T):
                masks[t, b] = O.contacts_to_mask(
                    id2obj, d['r%d_contacts' % r][b, t, :d['r%d_ncon' % r][b, t]])
        out, envs = O.generate_rollouts(env_id, B, T, masks,
                                        num_grey=d['r%d_num_objs' % r] - 4, envs=envs)
        np.testing.assert_array_equal(out['ag'], d['r%d_ag' % r])
        np.testing.assert_array_equal(out['g'], d['r%d_g' % r])
        np.testing.assert_array_equal(out['info_is_success'], d['r%d_info_is_success' % r])
        assert out['success_rate'] == float(d['r%d_success_rate' % r])


@pytest.mark.parametrize('name', ['BlocksTouch', 'BlocksTouchVariation', 'ToppleTower', 'GripperTouch', 'BlocksTouchChoose'])
def test_pg_discounted_return(name):
    """policy_gradient/rollout.py:255-258 return accumulation, bit-exact f64."""
    env_id = name + '-v0'
    v = VARIANTS[env_id]
    d = np.load(os.path.join(G, 'pg_rollout_%s.npz' % name))
    B, T = int(d['B']), int(d['T'])
    gamma = 1. - 1. / T                      # config.py:83
    envs = None
    for r in range(len(d['modes'])):
        masks = np.zeros((T, B), np.uint32)
        for b in range(B):
            id2obj = v['models'][int(d['r%d_model_idx' % r][b])]['id2obj']
            for t in range(T):
                masks[t, b] = O.contacts_to_mask(
                    id2obj, d['r%d_contacts' % r][b, t, :d['r%d_ncon' % r][b, t]])
        out, envs = O.generate_rollouts(env_id, B, T, masks,
                                        num_grey=d['r%d_num_objs' % r] - 4, envs=envs)
        for b in range(B):
            Gb = O.discounted_return(out['reward'][:, b].astype(np.float64), gamma)
            np.testing.assert_array_equal(Gb.view(np.uint64), d['r%d_G' % r][b].view(np.uint64))


def test_synth_mask_properties():
    """Build-defined contact source: deterministic, table-block pairs forced,
    no diagonal / out-of-range pairs, Bernoulli(5/32) elsewhere."""
    L = O.lib()
    m = O.synth_masks(7, np.arange(4096), 8, 6)
    assert (m == O.synth_masks(7, np.arange(4096), 8, 6)).all()
    forced = sum(1 << L.orc_pair_index(1, k) for k in range(2, 6))
    assert ((m & forced) == forced).all()
    diag = sum(1 << L.orc_pair_index(i, i) for i in range(6))
    assert ((m & diag) == 0).all()
    free = [L.orc_pair_index(i, j) for i in range(6) for j in range(i + 1, 6) if not (i == 1 and j >= 2)]
    rate = np.mean([((m >> p) & 1).mean() for p in free])
    assert abs(rate - 5 / 32) < 0.01
    m4 = O.synth_masks(7, np.arange(512), 4, 4)
    valid4 = sum(1 << L.orc_pair_index(i, j) for i in range(4) for j in range(i + 1, 4))
    assert ((m4 & ~np.uint32(valid4)) == 0).all()


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_observation_assembly(env_id):
    """Reference _get_obs (fetch_env.py:187-228, Variation :567-621) on seeded random site /
    robot values: layout, dt scaling, relative quantities, Variation one-hots + padding.
    Bit-exact float64.  (mat2euler is third-party: the fixture passes euler angles in.)"""
    d = np.load(os.path.join(G, 'obs_%s.npz' % short(env_id)))
    e = O.OracleEnv(env_id)
    assert d['obs'].shape[1] == VARIANTS[env_id]['dimo']
    for k in range(len(d['obs'])):
        e.e.num_objs = int(d['num_objs'][k])
        nb = e.e.num_objs - 2
        phys = {f: d[f][k] for f in ('grip_pos', 'grip_velp', 'gripper_state', 'gripper_vel')}
        phys.update({f: d[f][k][:nb] for f in ('obj_pos', 'obj_rot', 'obj_velp', 'obj_velr')})
        got = e.get_obs(phys, float(d['dt']))
        np.testing.assert_array_equal(got.view(np.uint64), d['obs'][k].view(np.uint64))


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_reset_placement(env_id):
    """Reference _randomize_objects (all five classes, train / set_test / challenge modes,
    curriculum levels, gripper near the table edge) replayed with the recorded standard
    variates: identical accept/reject decisions (same number of draws consumed) and
    bit-identical positions against the fixture (NumPy 2.2.6 of the build container; the
    2-vector norm goes through BLAS ddot, so other BLAS builds may differ by an ulp)."""
    d = np.load(os.path.join(G, 'placement_%s.npz' % short(env_id)))
    e = O.OracleEnv(env_id)
    assert len(d['xy']) >= 30
    for k in range(len(d['xy'])):
        e.e.num_objs = int(d['num_objs'][k])
        nb = e.e.num_objs - 2
        xy, used, fail = e.randomize_objects(
            d['grip_xy'][k], d['obj_range'][k], np.nan_to_num(d['max_obj_range'][k]),
            np.nan_to_num(d['wrong_obj_range'][k]), d['test'][k], d['challenge'][k],
            d['eu'][k, :d['n_eu'][k]], d['gz'][k, :d['n_gz'][k]], d['gu'][k, :d['n_gu'][k]])
        assert not fail
        assert used == [int(d['n_eu'][k]), int(d['n_gz'][k]), int(d['n_gu'][k])]
        np.testing.assert_array_equal(xy.view(np.uint64), d['xy'][k, :nb].view(np.uint64))
    # one variate short -> reports exhaustion instead of looping
    k = int(np.argmax(d['n_eu']))
    e.e.num_objs = int(d['num_objs'][k])
    _, _, fail = e.randomize_objects(d['grip_xy'][k], d['obj_range'][k], np.nan_to_num(d['max_obj_range'][k]),
                                     np.nan_to_num(d['wrong_obj_range'][k]), d['test'][k], d['challenge'][k],
                                     d['eu'][k, :d['n_eu'][k] - 1], d['gz'][k, :d['n_gz'][k]],
                                     d['gu'][k, :d['n_gu'][k]])
    assert fail or d['n_eu'][k] == 0


@pytest.mark.parametrize('env_id', ENV_IDS)
def test_set_action_vector(env_id):
    """robot_env.py:58 clip + fetch_env.py:170-185: the 9-vector the reference hands to the
    robotics utils (float32 scaling by 0.05, block_gripper zeroing), bit-exact."""
    d = np.load(os.path.join(G, 'obs_%s.npz' % short(env_id)))
    assert bool(d['block_gripper']) == VARIANTS[env_id]['block_gripper']
    assert (np.abs(d['actions']) > 1).any()
    for a, exp in zip(d['actions'], d['set_action']):
        np.testing.assert_array_equal(O.set_action(a, bool(d['block_gripper'])).view(np.uint64), exp.view(np.uint64))


def test_fixture_generator_is_deterministic():
    """tests/golden/make_golden.py --check: every committed fixture regenerates bit-identically
    from the reference (only where /root/reference is mounted: the build container)."""
    import subprocess
    import sys
    if not os.path.isdir('/root/reference/gym_blocks'):
        pytest.skip('the reference is not mounted here')
    out = subprocess.run([sys.executable, os.path.join(G, 'make_golden.py'), '--check'], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'regenerates bit-identically' in out.stdout


def test_compute_reward_summation_order_matches_numpy_and_the_reference_fixture():
    """fetch_env.py:141 sums achieved_goal * goal with np.sum: NumPy's pairwise order, in the arrays' precision.
    The oracle restates that order (third-party NumPy); here it is checked against NumPy itself on random rows
    and against the reference's compute_reward on rows whose outcome depends on the order
    (tests/golden/compute_reward_order.npz: every row's exact sum equals count_nonzero(g))."""
    rng = np.random.RandomState(3)
    for dt in (np.float32, np.float64):
        for n in (1, 5, 8, 9, 36, 100, 128, 129, 255, 256, 300):
            a = (rng.normal(size=(50, n)) * 10 ** rng.uniform(-4, 4, size=(50, n))).astype(dt)
            np.testing.assert_array_equal(O.rowsum(a), np.sum(a, axis=-1))
    d = np.load(os.path.join(G, 'compute_reward_order.npz'))
    flips = 0
    for key in sorted(k[:-3] for k in d.files if k.endswith('_ag')):
        ag, g, want = d[key + '_ag'], d[key + '_g'], d[key + '_rbits']
        got = O.compute_reward_f32(ag, g) if ag.dtype == np.float32 else O.compute_reward(ag, g)
        np.testing.assert_array_equal(got.view(np.uint32), want, err_msg=key)
        flips += int(d[key + '_differs_from_sequential'])
    assert flips > 50          # the fixture does separate NumPy's order from a sequential sum


def test_env_key_mirrors_the_reference_seed_arithmetic():
    """rollout.py:206-210 `env.seed(seed + 1000 * idx)`: the oracle's statement of the per-env key is a function
    of seed + 1000 * idx alone (64-bit wrap included)."""
    L = O.lib()
    for s, i in ((5, 3), (2 ** 40 + 17, 123456), (2 ** 64 - 2500, 5)):
        assert L.orc_env_key(s, i, 0) == L.orc_env_key((s + 1000 * i) % 2 ** 64, 0, 0)
        assert L.orc_synth_mask(s, i, 7, 6) == L.orc_synth_mask((s + 1000 * i) % 2 ** 64, 0, 7, 6)
        assert L.orc_num_grey_draw(s, i, 2) == L.orc_num_grey_draw((s + 1000 * i) % 2 ** 64, 0, 2)
    assert L.orc_env_key(5, 3, 0) != L.orc_env_key(5, 3, 1) and L.orc_env_key(5, 3, 0) != L.orc_env_key(6, 3, 0)


# ------------------------------------------------------------------ expert input trimming (pg worker)
TRIM_CASES = [('variation', True), ('choose', False), ('tower', False), ('same', False)]


@pytest.mark.parametrize('name,is_variation', TRIM_CASES)
def test_trim_restatement_equals_the_references_trim(name, is_variation):
    """policy_gradient/rollout.py:105-171 (batched branch): values AND dtypes (Variation returns float64)."""
    d = np.load(os.path.join(G, 'trim.npz'))
    dimo, dimg, num_objs = (int(x) for x in d[name + '_dims'])
    o, g, ag = d[name + '_o'], d[name + '_g'], d[name + '_ag']
    o_, g_, ag_ = O.trim(o, g, ag, dimo, dimg, is_variation, num_objs=num_objs)
    for got, key in ((o_, 'o_out'), (g_, 'g_out'), (ag_, 'ag_out')):
        want = d['%s_%s' % (name, key)]
        assert got.dtype == want.dtype and got.shape == want.shape, key
        np.testing.assert_array_equal(got, want, err_msg=key)
    if name == 'same':                               # "No need to trim": the very same objects
        assert d['same_same_objects'].all() and o_ is o and g_ is g and ag_ is ag
    else:
        assert not d[name + '_same_objects'].any()


def test_trim_restatement_on_the_inputs_the_reference_expert_saw():
    """The expert of the reference's pg rollouts was handed trim(o_t, g, ag_t): the recorded inputs equal the
    restatement applied to the episode arrays of the same fixture, at every step the expert was in charge."""
    for name, is_variation in (('BlocksTouchVariation', True), ('BlocksTouchChoose', False)):
        d = np.load(os.path.join(G, 'pg_expert_rollout_%s.npz' % name))
        dimo, dimg = (int(x) for x in d['expert_dims'])
        for r in range(int(d['n_rollouts'])):
            steps = np.flatnonzero(d['r%d_expert_in_charge' % r])
            assert len(steps) == d['r%d_expert_saw_o' % r].shape[0] and len(steps) > 0
            for k, t in enumerate(steps):
                o_, g_, ag_ = O.trim(d['r%d_o' % r][:, t], d['r%d_g' % r][:, t], d['r%d_ag' % r][:, t], dimo, dimg, is_variation)
                for got, key in ((o_, 'o'), (g_, 'g'), (ag_, 'ag')):
                    want = d['r%d_expert_saw_%s' % (r, key)][k]
                    assert got.dtype == want.dtype
                    np.testing.assert_array_equal(got, want)


def test_trim_raises_as_numpy_does_when_a_row_has_the_wrong_number_of_coloured_blocks():
    d = np.load(os.path.join(G, 'trim.npz'))
    o = d['variation_o'].copy()
    o[3, 11 + 15:11 + 19] = [1, 0, 0, 0]             # slot 0 of row 3 painted grey: one coloured block left
    with pytest.raises(ValueError):
        O.trim(o, d['variation_g'], d['variation_ag'], 40, 16, True)
    with pytest.raises(AssertionError):              # 3 objects of a 6 x 6 goal are 9 entries, not 16
        O.trim(d['variation_o'], d['variation_g'], d['variation_ag'], 40, 16, True, num_objs=3)


@pytest.mark.parametrize('name,pg', [('BlocksTouch', False), ('BlocksTouchVariation', False), ('ToppleTower', False),
                                     ('BlocksTouchChooseCurriculum', False), ('GripperTouch', False),
                                     ('BlocksTouch', True), ('BlocksTouchVariation', True)])
def test_restart_fixtures_replayed_on_the_oracle_envs(name, pg):
    """rollout.py:136-142 / policy_gradient/rollout.py:243-246 around a scripted NaN or MujocoException, on the oracle's
    matrix-form envs: every env takes the steps it took in the reference's discarded attempt (all envs k + 1 after a NaN at step
    k; envs 0 .. j-1 k + 1 and the rest k when env j raised), is reset as often as the reference reset it (three times after a
    NaN, twice after an exception), then runs the T steps of the attempt the reference returned.  With quirk Q1 the base
    variants keep what the discarded steps touched, so the returned episode depends on exactly where the attempt ended."""
    env_id = name + '-v0'
    v = VARIANTS[env_id]
    d = np.load(os.path.join(G, ('pg_restart_%s.npz' if pg else 'restart_%s.npz') % name))
    B, T = int(d['B']), int(d['T'])
    id2objs = [m['id2obj'] for m in v['models']]
    envs = [O.OracleEnv(env_id) for _ in range(B)]
    for e in envs:
        e.reset(0)                                   # RolloutStudent.__init__ resets every env once (rollout.py:44)
    for r in range(int(d['n_calls'])):
        kind, j, k = (int(x) for x in d['faults'][r])
        nobj = d['r%d_reset_num_objs' % r]           # (n_resets, B)
        assert nobj.shape[0] == {0: 1, 1: 3, 2: 2}[kind]
        nd = d['r%d_discard_steps' % r]
        want_nd = [0] * B if kind == 0 else ([k + 1] * B if kind == 1 else [k + 1 if b < j else k for b in range(B)])
        assert nd.tolist() == want_nd
        ag = np.zeros((B, T + 1, envs[0].dimg))
        succ = np.zeros((B, T, 1), np.float32)
        rew = np.zeros((T, B), np.float32)
        for b, e in enumerate(envs):
            if kind:
                e.reset(int(nobj[0, b]) - 4)
                tab = id2objs[int(nobj[0, b]) - 4 if len(id2objs) > 1 else 0]
                for t in range(int(nd[b])):
                    e.step(tab, d['r%d_discard_contacts' % r][b, t, :d['r%d_discard_ncon' % r][b, t]])
                for extra in range(1, nobj.shape[0] - 1):          # reset_all_rollouts before the retry (NaN only)
                    e.reset(int(nobj[extra, b]) - 4)
            ag[b, 0] = e.reset(int(nobj[-1, b]) - 4)
            tab = id2objs[int(d['r%d_model_idx' % r][b])]
            for t in range(T):
                a, rr, _, s_ = e.step(tab, d['r%d_contacts' % r][b, t, :d['r%d_ncon' % r][b, t]])
                ag[b, t + 1], succ[b, t, 0], rew[t, b] = a, s_, rr
        np.testing.assert_array_equal(ag.astype(np.float32), d['r%d_ag' % r], err_msg='call %d' % r)
        np.testing.assert_array_equal(succ, d['r%d_info_is_success' % r])
        assert float(np.mean(succ[:, -1, 0].astype(np.float64))) == float(d['r%d_success_rate' % r])      # np.mean of a float64 array (rollout.py:123,165)
        np.testing.assert_array_equal(np.stack([e.achieved_goal() for e in envs]), d['r%d_final_ag' % r])
        if pg:
            for b in range(B):
                Gb = O.discounted_return(rew[:, b].astype(np.float64), 1. - 1. / T)
                np.testing.assert_array_equal(Gb.view(np.uint64), d['r%d_G' % r][b].view(np.uint64))

"""Env registry mirroring gym_blocks/__init__.py:6-53: the same seven ids, every
one with max_episode_steps=50 and kwargs {'reward_type': 'sparse'}.

`make(id)` plays the role of `gym.make(id)`: it returns an env that already
carries the TimeLimit behaviour (`_max_episode_steps`, `done`), because gym's
wrapper is what the reference's callers see (config.py:75-80).  If the real
`gym` package is importable the ids are also registered there under the same
names (entry point = this package), so `gym.make('BlocksTouch-v0')` works.
"""
from collections import OrderedDict


class EnvSpec:
    def __init__(self, id, entry_point, kwargs, max_episode_steps):
        self.id, self.entry_point = id, entry_point
        self.kwargs, self.max_episode_steps = dict(kwargs), max_episode_steps


registry = OrderedDict()


def register(id, entry_point, kwargs=None, max_episode_steps=None):
    registry[id] = EnvSpec(id, entry_point, kwargs or {}, max_episode_steps)


for _id, _cls in [
        ('GripperTouch-v0', 'GripperTouchEnv'),
        ('BlocksTouch-v0', 'BlocksTouchEnv'),
        ('ToppleTower-v0', 'ToppleTowerEnv'),
        ('BlocksTouchCurriculum-v0', 'BlocksTouchCurriculum'),
        ('BlocksTouchChoose-v0', 'BlocksTouchChooseEnv'),
        ('BlocksTouchChooseCurriculum-v0', 'BlocksTouchChooseCurriculum'),
        ('BlocksTouchVariation-v0', 'BlocksTouchVariationEnv')]:
    register(id=_id, entry_point='gym_blocks.envs:' + _cls,
             kwargs={'reward_type': 'sparse'}, max_episode_steps=50)


def make(env_id, num_envs=None, **kwargs):
    """gym.make(env_id) for the batched backend.

    num_envs=None -> a drop-in single env: 1-D numpy observations with the
    reference's dtypes, scalar reward/done, python-bool info (what config.py and
    rollout.py expect from one env).  num_envs=B -> batched torch tensors."""
    from .envs import BatchedBlocksEnv
    if env_id not in registry:
        raise KeyError('No registered env with id: %s' % env_id)
    spec = registry[env_id]
    kw = dict(spec.kwargs)
    kw.update(kwargs)
    kw.setdefault('max_episode_steps', spec.max_episode_steps)
    if num_envs is None:
        return BatchedBlocksEnv(env_id, num_envs=1, scalar_api=True, **kw)
    return BatchedBlocksEnv(env_id, num_envs=num_envs, **kw)


def _gym_entry(env_id):
    def entry(**kwargs):
        from .envs import BatchedBlocksEnv
        kwargs.pop('reward_type', None)
        # gym.make wraps the env in its own TimeLimit (max_episode_steps below), which then
        # agrees with the env's built-in step counter
        return BatchedBlocksEnv(env_id, num_envs=1, scalar_api=True, **kwargs)
    return entry


def register_with_gym(prefix=''):
    """Optional: if the real `gym` package is installed, register the seven ids with it
    (same names, max_episode_steps=50, kwargs={'reward_type': 'sparse'} as
    gym_blocks/__init__.py:6-53), so `gym.make('BlocksTouch-v0')` returns the drop-in single
    env.  gym is not a dependency and is absent from the build image: the call is tested against
    a recording stand-in for gym.envs.registration (tests/test_abi_cpu.py); returns the list of ids
    registered (empty if gym is missing)."""
    try:
        from gym.envs.registration import register as gym_register
    except Exception:  # noqa: BLE001
        return []
    done = []
    for spec in registry.values():
        gym_register(id=prefix + spec.id, entry_point=_gym_entry(spec.id), kwargs=dict(spec.kwargs),
                     max_episode_steps=spec.max_episode_steps)
        done.append(prefix + spec.id)
    return done

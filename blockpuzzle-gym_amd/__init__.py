"""blockpuzzle-gym_amd: MI355X-native batched gym_blocks step()/reset() path.

Import as `blockpuzzle_gym_amd` (the repo-root shim maps the importable name
onto this hyphenated directory).  Public surface mirrors the reference:
  make / register / registry      gym_blocks/__init__.py + gym.make
  BatchedBlocksEnv                gym_blocks/envs (robot_env.py, fetch_env.py)
  RolloutStudent                  gym_blocks/rollout.py
  PGRolloutStudent                gym_blocks/policy_gradient/rollout.py (return G)
"""
from . import _lib
from .registration import make, register, registry, register_with_gym
from .envs import (BatchedBlocksEnv, SyntheticContacts, ScriptedContacts, CallableContacts, PhysicsError,
                   compute_reward, mat2euler, geom2objid, default_geom_names)
from .rollout import (RolloutStudent, PGRolloutStudent, configure_dims, discounted_return, discounted_return_scan,
                      time_major_empty)
from . import distributed

__all__ = ['make', 'register', 'registry', 'register_with_gym', 'BatchedBlocksEnv', 'RolloutStudent', 'PGRolloutStudent',
           'SyntheticContacts', 'ScriptedContacts', 'CallableContacts', 'PhysicsError', 'compute_reward',
           'configure_dims', 'discounted_return', 'discounted_return_scan', 'time_major_empty', 'mat2euler', 'geom2objid', 'default_geom_names', 'distributed']

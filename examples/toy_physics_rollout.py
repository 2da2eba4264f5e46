"""A physics backend plugged into the contacts-in boundary, end to end (needs one MI355X).

The reference calls MuJoCo between `_set_action` and `_step_callback` (robot_env.py:59-61).  Here a
TOY kinematic stand-in (device tensors, ~40 lines: a gripper that moves by the clipped control, cubes
that are pushed when the gripper runs into them) plays that part -- it is NOT MuJoCo and not part of
the product; it shows what a backend has to provide and how the pieces meet:

  reset()            -> bpg_randomize_objects places the cubes (fetch_env.py:370-399), on_reset hands
                        the positions to the backend
  policy             -> a scripted pusher: go behind the green cube, push it onto the blue one
  step()             -> backend.fn(env, actions): env.set_action_vector (clip + _set_action), moves
                        things, emits geom-id contact lists as sim.data.contact would, and writes the
                        observation through env.assemble_observation
  rollout worker     -> RolloutStudent.generate_rollouts: episode arrays, success statistics

    python examples/toy_physics_rollout.py [num_envs]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blockpuzzle_gym_amd as bpg

ENV_ID = 'BlocksTouch-v0'
HALF = 0.025                      # cube half size (assets/fetch/2blocks.xml)
TABLE_Z = 0.425


class ToyPhysics:
    """Kinematic gripper + pushable cubes; contacts by proximity."""

    def __init__(self, B, device, geom_names):
        self.B, self.dev = B, device
        self.finger = next(i for i, n in enumerate(geom_names) if n and 'finger' in n)
        self.table = geom_names.index('table')
        self.cube = [geom_names.index('object%d' % k) for k in range(2)]
        self.grip = torch.zeros(B, 3, dtype=torch.float64, device=device)
        self.grip_v = torch.zeros_like(self.grip)
        self.xy = torch.zeros(B, 2, 2, dtype=torch.float64, device=device)
        self.vel = torch.zeros(B, 2, 3, dtype=torch.float64, device=device)
        self.zeros2 = torch.zeros(B, 2, dtype=torch.float64, device=device)
        self.zeros23 = torch.zeros(B, 2, 3, dtype=torch.float64, device=device)

    # -- env.reset() / env.set_test() ------------------------------------------------------------
    def on_reset(self, env, xy, status, env_mask, test):
        sel = slice(None) if env_mask is None else env_mask.bool()
        self.xy[sel] = xy[sel]
        self.grip[sel] = torch.tensor([env.grip_xy[0], env.grip_xy[1], TABLE_Z + 0.1], dtype=torch.float64, device=self.dev)
        self.vel[sel] = 0
        self._write_observation(env)

    # -- env.step(): between _set_action and _step_callback ---------------------------------------
    def __call__(self, env, actions):
        ctrl = env.set_action_vector(actions)                    # (B, 9): clipped pos delta, quaternion, fingers
        new = self.grip + ctrl[:, :3]
        new[:, 2] = new[:, 2].clamp(TABLE_Z + HALF, TABLE_Z + 0.3)
        self.grip_v = (new - self.grip) / 0.04
        low = new[:, 2] < TABLE_Z + 3 * HALF                     # only a lowered gripper meets the cubes
        old_xy = self.xy.clone()
        for k in range(2):                                       # a cube in the gripper's way is carried along
            d = self.xy[:, k] - new[:, :2]
            dist = d.norm(dim=1)
            hit = low & (dist < 2 * HALF)
            push = torch.where(hit[:, None], d / dist.clamp_min(1e-9)[:, None] * (2 * HALF - dist)[:, None], self.zeros2)
            self.xy[:, k] += push
        apart = self.xy[:, 0] - self.xy[:, 1]                    # cubes do not overlap
        gap = apart.norm(dim=1)
        over = (2 * HALF - gap).clamp_min(0)
        self.xy[:, 0] += apart / gap.clamp_min(1e-9)[:, None] * over[:, None] * 0.5
        self.xy[:, 1] -= apart / gap.clamp_min(1e-9)[:, None] * over[:, None] * 0.5
        self.vel[:, :, :2] = (self.xy - old_xy) / 0.04
        self.grip = new
        self._write_observation(env)
        # what sim.data.contact would list: cube-table always, finger-cube / cube-cube by proximity
        B = self.B
        pairs = torch.zeros(B, 8, 2, dtype=torch.int16, device=self.dev)
        ncon = torch.zeros(B, dtype=torch.int32, device=self.dev)

        def add(mask, g1, g2):
            idx = ncon.long()
            rows = torch.nonzero(mask)[:, 0]
            pairs[rows, idx[rows], 0] = g1
            pairs[rows, idx[rows], 1] = g2
            ncon[rows] += 1
        everyone = torch.ones(B, dtype=torch.bool, device=self.dev)
        for k in range(2):
            add(everyone, self.table, self.cube[k])
            add(low & ((self.xy[:, k] - self.grip[:, :2]).norm(dim=1) < 2 * HALF + 2e-3), self.finger, self.cube[k])
        add((self.xy[:, 0] - self.xy[:, 1]).norm(dim=1) < 2 * HALF + 2e-3, self.cube[0], self.cube[1])
        return pairs, ncon

    def _write_observation(self, env):
        z = torch.full((self.B, 2, 1), TABLE_Z, dtype=torch.float64, device=self.dev)
        env.assemble_observation(dict(grip_pos=self.grip, grip_velp=self.grip_v, gripper_state=self.zeros2,
                                      gripper_vel=self.zeros2, obj_pos=torch.cat([self.xy, z], 2), obj_rot=self.zeros23,
                                      obj_velp=self.vel, obj_velr=self.zeros23))


class PusherPolicy:
    """Scripted: lower the gripper behind the green cube (object0) as seen from the blue one
    (object1), then push along that line.  Reads only the observation (fetch_env.py:187-228 layout:
    grip_pos 0:3, block k at 10+15k: pos, pos-grip, ...)."""
    device_tensors = True

    def get_actions(self, o, ag, g, **kw):
        grip, green, blue = o[:, 0:3], o[:, 10:13], o[:, 25:28]
        line = green[:, :2] - blue[:, :2]
        line = line / line.norm(dim=1, keepdim=True).clamp_min(1e-6)
        behind = green[:, :2] + line * 0.07
        at_height = grip[:, 2] < TABLE_Z + 0.06
        near_line = (grip[:, :2] - behind).norm(dim=1) < 0.015
        rel = grip[:, :2] - green[:, :2]
        along = (rel * line).sum(1, keepdim=True)
        lateral = (rel - along * line).norm(dim=1)
        pushing = at_height & (along[:, 0] > 0.0) & (lateral < 0.02)
        u = torch.zeros(o.shape[0], 4, device=o.device)
        # phase 1: fly over to the point behind the green cube, staying high
        to = (behind - grip[:, :2]) / 0.05
        u[:, :2] = torch.where((near_line | pushing)[:, None], torch.zeros_like(to), to)
        u[:, 2] = torch.where(near_line | pushing, torch.full_like(u[:, 2], -1.0), (TABLE_Z + 0.1 - grip[:, 2]) / 0.05)
        # phase 2: low enough -> push towards the blue cube
        u[:, :2] = torch.where(pushing[:, None], -line * 0.4, u[:, :2])
        u[:, 2] = torch.where(pushing, torch.zeros_like(u[:, 2]), u[:, 2])
        return u


def run(B=4096, device='cuda:0', seed=0, test=False):
    names = bpg.default_geom_names(ENV_ID)[0]
    phys = ToyPhysics(B, torch.device(device), names)
    source = bpg.CallableContacts(phys, on_reset=phys.on_reset, writes_observation=True)
    worker = bpg.RolloutStudent(lambda num_envs: bpg.make(ENV_ID, num_envs=num_envs, device=device, seed=seed,
                                                          contact_source=source),
                                PusherPolicy(), dict(o=40, u=4, g=16, info_is_success=1), None, 50, rollout_batch_size=B)
    episode = worker.generate_rollouts(test=test)
    return worker, episode, phys


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    worker, ep, phys = run(B)
    touch = ep['ag'].reshape(B, 51, 4, 4)[:, :, 2, 3]
    print('%d envs x 50 steps with the toy backend: success rate %.3f (cubes touching at the end in %.3f of the envs)'
          % (B, worker.success_history[-1], (touch[:, -1] == 1).float().mean().item()))

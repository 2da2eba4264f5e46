#!/usr/bin/env python3
"""Prints the markdown table of profiles/README.md from profiles/<round>_pmc/kernel_means.json and
traffic_all_kernels.json (tools/publish_profiles.sh writes both):  python tools/profile_table.py [r05]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else 'r05'
ROWS = [
    ('k_rollout_masks_bench', '`k_rollout<masks>` (headline, 2/lane × 1024), as `bench.py` issues it', 'reset + 50 steps + success count, 1 M envs'),
    ('k_rollout_masks', 'same, from `prof_kernels.py` (its own process and buffers)', ''),
    ('k_rollout_masks_with_episode_return', 'same + the per-env episode return (`d_episode_return`, 4 B/env)', ''),
    ('k_rollout_masks_524288', 'same, 524 288 envs', ''),
    ('k_rollout_masks_524288_with_episode_return', 'same, 524 288 envs + the episode return (configs[3] per GPU)', ''),
    ('k_rollout_masks_262144', 'same, 262 144 envs (2/lane × 256)', ''),
    ('k_rollout_masks_131072', 'same, 131 072 envs (1/lane, 5 rows in flight)', ''),
    ('k_rollout_masks_65536', 'same, 65 536 envs (1/lane, 10 rows in flight)', ''),
    ('k_rollout_masks_4194304_aliased_rows', 'same, 4 194 304 envs, rows a power of two apart (2/lane × 256)', ''),
    ('k_rollout_masks_x10', '`k_rollout<masks, 10 rollouts per launch>`, 1 M envs', '10 × (reset + 50 steps + count) in one launch'),
    ('k_rollout_masks_65536_x10', 'same, 65 536 envs', ''),
    ('k_rollout_synth', '`k_rollout<synth>`', 'reset + 50 steps, contacts generated in-kernel'),
    ('k_rollout_masks_flags_okbits', '`k_rollout<masks, flags + step-ok words>`', 'pg path, first launch of `bpg_rollout(d_G)`'),
    ('k_returns_from_okbits_lane50', '`k_returns_from_okbits_lane<50>`', 'pg path, second launch: G rows from the step-ok words'),
    ('k_discounted_return_lane50_exact', '`k_discounted_return_lane<50, exact>`', 'pg return from a reward stream, bit-exact'),
    ('k_discounted_return_lane50_scan', '`k_discounted_return_lane<50, scan>`', 'the same as an O(T) scan'),
    ('k_step_vec4_masks', '`k_step_vec4<masks>`', 'one `env.step`, HBM-cold (13 batches round-robin)'),
    ('k_step_record', '`k_step_record<6>`', 'step + f32 `achieved_goal` row + f32 `is_success`'),
    ('k_episode', '`k_episode<6>`', 'reset + 50 steps + float32 episode arrays'),
    ('k_trim_variation', '`k_trim<variation>`', "the expert's inputs of one pg step (87 → 40 f64, 2 × 36 → 16)"),
    ('k_step_contacts', '`k_step_contacts<4>`', 'one step from geom-id pair lists (maxcon 16), lists HBM-cold'),
    ('k_contacts_to_masks_tb', '`k_contacts_to_masks<4>`', 'a recorded rollout\'s lists (T = 50, 262 144 envs, maxcon 16) → mask stream'),
    ('k_compute_reward_f32', '`k_compute_reward_f32`', "4 M rows of 36 floats, NumPy's summation order"),
    ('k_compute_reward_f32_goal', '`k_compute_reward_f32<one goal row>`', '4 M rows of 36 floats against one goal row'),
    ('k_compute_reward_f64', '`k_compute_reward_f64`', "2 M rows of 36 doubles (the precision of `env.step`'s own call)"),
    ('k_compute_reward_f64_goal', '`k_compute_reward_f64<one goal row>`', '2 M rows of 36 doubles against one goal row'),
    ('k_assemble_obs_f32', '`k_assemble_obs<float, 4>`', '`_get_obs` rows of 1 M Variation envs'),
    ('k_nan_flag', '`k_nan_flag`', 'NaN check of 1 M × 87 floats'),
    ('k_set_action', '`k_set_action`', 'clip + 9-vector, 1 M envs'),
]


def size(b):
    return '%.2f GB' % (b / 1e9) if b >= 1e9 else '%.1f MB' % (b / 1e6)


def main():
    m = json.load(open(os.path.join(ROOT, 'profiles', R + '_pmc', 'kernel_means.json')))
    t = json.load(open(os.path.join(ROOT, 'profiles', R + '_pmc', 'traffic_all_kernels.json')))
    print('| kernel | launch covers | algorithmic bytes | mean (calls) | of 8 TB/s | PMC traffic (read / write) |')
    print('|---|---|---|---|---|---|')
    for key, name, covers in ROWS:
        if key not in m:
            continue
        v, tt = m[key], t.get(key, {})
        us = v['mean_us']
        dur = '%.2f ms' % (us / 1e3) if us >= 1000 else '%.1f µs' % us
        frac = v.get('frac_of_8TBs')
        traffic = '%.1f / %.1f MB' % (tt['hbm_read_bytes'] / 1e6, tt['hbm_write_bytes'] / 1e6) if tt else ''
        print('| %s | %s | %s | %s (%d) | %s | %s |' % (name, covers, size(v['algorithmic_bytes']), dur, v['calls'],
                                                       ('%.3f' % frac) if frac else '', traffic))


if __name__ == '__main__':
    main()

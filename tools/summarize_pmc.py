#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into per-kernel HBM bytes.

    python tools/summarize_pmc.py <dir with fetch_*/ and write_*/ runs> <out dir>

Expects (tools/collect_profiles.sh writes them):
    cal_fetch/  cal_write/      tools/pmc_calibrate           -> bytes-per-count for each access width
    k_fetch/    k_write/        tools/prof_kernels.py ...     -> every kernel of the path
    b_fetch/    b_write/        bench.py --steps 8 ...        -> the headline launch as the bench issues it
Counters are in KiB.  gfx950 under-reports wide coalesced reads (MI355X_MICROARCH.md, HBM section), so
every kernel's FETCH_SIZE is multiplied by the factor the calibration kernel of the same access
width shows in the SAME session; WRITE_SIZE likewise (it has been exact so far)."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict

GIB = 1 << 30
# calibration kernel name fragment -> (known bytes, counter)
CAL = {
    'cal_read<HIP_vector_type<unsigned int, 4': (GIB, 'FETCH_SIZE', 'read16'),
    'cal_read<HIP_vector_type<unsigned int, 2': (GIB, 'FETCH_SIZE', 'read8'),
    'cal_read<unsigned int>': (GIB, 'FETCH_SIZE', 'read4'),
    'cal_read_u8': (GIB // 4, 'FETCH_SIZE', 'read1'),
    'cal_write<HIP_vector_type<unsigned int, 4u>, true': (GIB, 'WRITE_SIZE', 'write16nt'),
    'cal_write<HIP_vector_type<unsigned int, 4u>, false': (GIB, 'WRITE_SIZE', 'write16'),
    'cal_write<HIP_vector_type<unsigned int, 2u>, true': (GIB, 'WRITE_SIZE', 'write8nt'),
    'cal_write<HIP_vector_type<unsigned int, 2u>, false': (GIB, 'WRITE_SIZE', 'write8'),
    'cal_write<unsigned int, true': (GIB, 'WRITE_SIZE', 'write4nt'),
    'cal_write<unsigned short, true': (GIB // 4, 'WRITE_SIZE', 'write2nt'),
    'cal_write<unsigned char, true': (GIB // 8, 'WRITE_SIZE', 'write1nt'),
    'cal_write<double, false': (GIB, 'WRITE_SIZE', 'write8f64'),
}
# (kernel name fragment, grid size in threads or None, label, read width key, algorithmic read bytes, algorithmic
#  write bytes, collection group).  The grid size tells launch shapes of one instantiation apart; the group is the
# tools/collect_profiles.sh invocation the launch belongs to ('' = first kernel group / bench, '2' = second group).
B, T = 1 << 20, 50


def roll(n):
    return n * (4 * T + 4), n * (5 * T + 8)


# template arguments: <SRC, EPL, OUT, MODE (0 continue / 1 reset / 2 several rollouts), UNROLL, PIPE, LB, AUX (0 / 1 step-ok words / 2 per-env counter)>
FAT = 'k_rollout<0, 2, 3, 1, 5, false, 1024, 0>'
FAT_RET = 'k_rollout<0, 2, 3, 1, 5, false, 1024, 2>'      # + the per-env episode return
MID = 'k_rollout<0, 2, 3, 1, 5, false, 256, 0>'


def roll_ret(n):
    r, w = roll(n)
    return r, w + 4 * n
K_MULTI = 10        # rollouts per launch in tools/prof_kernels.py multi64k / multi1m


def roll_k(n):
    """K rollouts in one launch: what K launches move (explicit num_grey row + masks in, streams + state out, per rollout)."""
    r, w = roll(n)
    return K_MULTI * r, K_MULTI * w


KERNELS = [
    (FAT, 256 * 1024, 'k_rollout_masks', 'read8') + roll(B) + ('',),
    (FAT, 256 * 1024, 'k_rollout_masks_524288', 'read8') + roll(524288) + ('2',),
    (FAT_RET, 256 * 1024, 'k_rollout_masks_with_episode_return', 'read8') + roll_ret(B) + ('',),
    (FAT_RET, 256 * 1024, 'k_rollout_masks_524288_with_episode_return', 'read8') + roll_ret(524288) + ('2',),
    (MID, 262144 // 2, 'k_rollout_masks_262144', 'read8') + roll(262144) + ('',),
    (MID, 1024 * 256, 'k_rollout_masks_4194304_aliased_rows', 'read8') + roll(4194304) + ('2',),
    ('k_rollout<0, 1, 3, 1, 5, true, 256, 0>', 131072, 'k_rollout_masks_131072', 'read4') + roll(131072) + ('',),
    ('k_rollout<0, 1, 3, 1, 10, true, 256, 0>', 65536, 'k_rollout_masks_65536', 'read4') + roll(65536) + ('',),
    ('k_rollout<1, 2, 3, 1, 10, false, 256, 0>', None, 'k_rollout_synth', 'read8', 0, B * (5 * T + 8), ''),
    ('k_rollout<0, 2, 2, 1, 5, false, 1024, 1>', None, 'k_rollout_masks_flags_okbits', 'read8',
     B * (4 * T + 4), B * (T + 8 + 8), ''),
    ('k_returns_from_okbits_lane<50', None, 'k_returns_from_okbits_lane50', 'read8', 8 * B, 8 * T * B, ''),
    ('k_step_vec4<0>', None, 'k_step_vec4_masks', 'read16', 12 * B, 13 * B, ''),
    ('k_step_record<6, 0', None, 'k_step_record', 'read4', 12 * B, (13 + 4 + 144) * B, ''),
    ('k_discounted_return_lane<50, 0', None, 'k_discounted_return_lane50_exact', 'read4', 4 * T * B, 8 * T * B, ''),
    ('k_discounted_return_lane<50, 1', None, 'k_discounted_return_lane50_scan', 'read4', 4 * T * B, 8 * T * B, ''),
    ('k_episode<6, 0', None, 'k_episode', 'read4', B * (4 * T + 8), B * (8 + 4 * (36 * (2 * T + 1) + T)), ''),
    ('k_compute_reward_f32<true, false>', None, 'k_compute_reward_f32', 'read16', 4 * B * 288, 4 * B * 4, ''),
    ('k_compute_reward_f32<true, true>', None, 'k_compute_reward_f32_goal', 'read16', 4 * B * 144, 4 * B * 4, ''),
    ('k_compute_reward_f64<true, false>', None, 'k_compute_reward_f64', 'read16', 2 * B * 576, 2 * B * 4, ''),
    ('k_compute_reward_f64<true, true>', None, 'k_compute_reward_f64_goal', 'read16', 2 * B * 288, 2 * B * 4, ''),
    ('k_reset<true>', None, 'k_reset_obs', 'read4', 8 * B, None, ''),
    ('k_assemble_obs<float, 4>', None, 'k_assemble_obs_f32', 'read8', B * (4 + 8 * 58), B * 4 * 87, ''),
    ('k_set_action', None, 'k_set_action', 'read16', 16 * B, 72 * B, ''),
    ('k_nan_flag<true>', None, 'k_nan_flag', 'read16', B * 4 * 87, 0, ''),
    ('k_trim<true>', None, 'k_trim_variation', 'read16', B * 4 * (87 + 2 * 36), B * (8 * 40 + 2 * 4 * 16), ''),
    ('k_step_contacts<4>', None, 'k_step_contacts', 'read16', B * (8 + 4 + 64), 13 * B, ''),
    ('k_contacts_to_masks<4>', None, 'k_contacts_to_masks_tb', 'read16', T * 262144 * (64 + 4) + 4 * 262144, 4 * T * 262144, ''),
    ('k_rollout<0, 1, 3, 2, 10, true, 256, 0>', 65536, 'k_rollout_masks_65536_x10', 'read4') + roll_k(65536) + ('',),
    ('k_rollout<0, 2, 3, 2, 5, false, 1024, 0>', 256 * 1024, 'k_rollout_masks_x10', 'read8') + roll_k(B) + ('2',),
]


def load(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel(rows, counter):
    """{(kernel name, grid size in threads): [bytes per dispatch]}"""
    out = defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == counter:
            out[(r['Kernel_Name'], int(r['Grid_Size']))].append(float(r['Counter_Value']) * 1024.0)
    return out


def pick(table, frag, grid):
    return [v for (name, g), vs in table.items() if frag in name and (grid is None or g == grid) for v in vs]


def trace_means(src, sub):
    """{(kernel name, grid threads): (calls, mean us, min us)} from the per-dispatch kernel trace."""
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            acc[(r['Kernel_Name'], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return acc


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    cal = {}
    for counter, sub in (('FETCH_SIZE', 'cal_fetch'), ('WRITE_SIZE', 'cal_write')):
        pk = per_kernel(load(os.path.join(src, sub, '**', '*counter_collection.csv')), counter)
        for frag, (known, cn, key) in CAL.items():
            if cn != counter:
                continue
            vals = pick(pk, frag, None)
            if vals:
                cal[key] = {'known_bytes': known, 'counter_bytes_median': statistics.median(vals),
                            'factor': known / statistics.median(vals)}
    json.dump(cal, open(os.path.join(dst, 'calibration.json'), 'w'), indent=1)
    res = {}
    for tag, group, suffix in (('k', '', ''), ('k2', '2', ''), ('b', '', '_bench')):
        fetch = per_kernel(load(os.path.join(src, tag + '_fetch', '**', '*counter_collection.csv')), 'FETCH_SIZE')
        write = per_kernel(load(os.path.join(src, tag + '_write', '**', '*counter_collection.csv')), 'WRITE_SIZE')
        for frag, grid, label, rkey, aread, awrite, grp in KERNELS:
            if grp != group:
                continue
            fv, wv = pick(fetch, frag, grid), pick(write, frag, grid)
            if not fv or not wv:
                continue
            rf = cal.get(rkey, {}).get('factor', 2.0 if rkey == 'read16' else 1.0)
            res[label + suffix] = {
                'launches': [len(fv), len(wv)], 'read_width_calibration': rkey, 'read_factor': rf,
                'hbm_read_bytes': statistics.median(fv) * rf, 'hbm_write_bytes': statistics.median(wv),
                'hbm_bytes_per_launch': statistics.median(fv) * rf + statistics.median(wv),
                'algorithmic_read': aread, 'algorithmic_write': awrite}
    json.dump(res, open(os.path.join(dst, 'traffic_all_kernels.json'), 'w'), indent=1)
    head = res.get('k_rollout_masks_bench') or res.get('k_rollout_masks')
    if head:
        json.dump({'k_rollout_masks': dict(head, envs=B, T=T, source=os.path.join(dst, 'traffic_all_kernels.json'),
                                            method='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; '
                                                   'FETCH_SIZE scaled by the factor an 8-B-per-lane read stream of known '
                                                   'size shows in the same session (calibration.json beside this file)',
                                            algorithmic_bytes_per_launch=B * (9 * T + 12))},
                  open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
    # rocprofv3 --kernel-trace means of the same launches, per (kernel, grid) (what bench.py quotes beside its
    # HIP-event figures)
    means = {}
    for sub, group, suffix in (('trace_kernels', '', ''), ('trace_kernels2', '2', ''), ('trace_headline', '', '_bench')):
        tm = trace_means(src, sub)
        for frag, grid, label, _, aread, awrite, grp in KERNELS:
            if grp != group:
                continue
            d = [v for (name, g), vs in tm.items() if frag in name and (grid is None or g == grid) for v in vs]
            if d:
                alg = None if awrite is None else aread + awrite
                means[label + suffix] = {'calls': len(d), 'mean_us': sum(d) / len(d), 'min_us': min(d), 'algorithmic_bytes': alg,
                                         'frac_of_8TBs': None if alg is None else alg / (sum(d) / len(d) * 1e-6) / 8e12}
    json.dump(means, open(os.path.join(dst, 'kernel_means.json'), 'w'), indent=1)
    print(json.dumps(cal, indent=1))
    for k, v in res.items():
        print('%-40s read %8.1f MB (alg %s)  write %8.1f MB (alg %s)' % (
            k, v['hbm_read_bytes'] / 1e6, None if v['algorithmic_read'] is None else round(v['algorithmic_read'] / 1e6, 1),
            v['hbm_write_bytes'] / 1e6, None if v['algorithmic_write'] is None else round(v['algorithmic_write'] / 1e6, 1)))
    for k, v in means.items():
        print('%-40s %4d calls  mean %9.2f us  min %9.2f us  %s' % (k, v['calls'], v['mean_us'], v['min_us'],
                                                                  '' if v['frac_of_8TBs'] is None else '%.3f of 8 TB/s' % v['frac_of_8TBs']))


if __name__ == '__main__':
    main()

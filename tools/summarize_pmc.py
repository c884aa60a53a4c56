#!/usr/bin/env python
"""Turns the rocprofv3 counter passes of tools/collect_profiles.sh (gpurun_out/<tag>/pmc_*) into
profiles/<tag>_pmc_summary.json -- the per-launch HBM traffic bench.py reports as roofline.traffic -- and copies the
judged summaries (kernel statistics, bench lines, phase profiles) into profiles/.

    python tools/summarize_pmc.py [tag]          (run here, after the gpurun call has merged gpurun_out/<tag>/)

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE reads half of a coalesced stream
(/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section); counters come from separate --pmc passes."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
KERNEL = 'ppn_kernel<2, 0, 1>'     # K_STEP, W = 2, Newton
BATCH = 4096


def per_launch(path):
    acc, launches = collections.defaultdict(float), set()
    for r in csv.DictReader(open(path)):
        if KERNEL in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
            launches.add(r['Dispatch_Id'])
    return {c: v / len(launches) for c, v in acc.items()}, len(launches)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    f, nf = per_launch(os.path.join(src, 'pmc_fetch', 'p_counter_collection.csv'))
    w, nw = per_launch(os.path.join(src, 'pmc_write', 'p_counter_collection.csv'))
    q, nq = per_launch(os.path.join(src, 'pmc_sq', 'p_counter_collection.csv'))
    hbm = (2 * f['FETCH_SIZE'] + w['WRITE_SIZE']) * 1024
    sha = None      # hash of the library the passes ran (tools/collect_profiles.sh writes it): bench.py refuses another build's traffic
    if os.path.exists(os.path.join(src, 'libppn.sha256')):
        sha = open(os.path.join(src, 'libppn.sha256')).read().split()[0]
    calib = None    # WRITE_SIZE / FETCH_SIZE against kernels that move a known number of bytes (tools/ubench/hbm_counter_calib.hip)
    if os.path.exists(os.path.join(src, 'hbm_counter_calib.json')):
        calib = json.load(open(os.path.join(src, 'hbm_counter_calib.json')))
    out = {
        'library_sha256': sha,
        'counter_calibration': calib,
        'command': 'rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 20 --warmup 3 --headline-only '
                   '(one pass per counter group, tools/collect_profiles.sh)',
        'kernel': KERNEL + ' = K_STEP, W=2 (IEEE-118, 118 active buses), Newton flavour, batch %d environments per launch' % BATCH,
        'launches_averaged': {'fetch': nf, 'write': nw, 'sq': nq},
        'FETCH_SIZE_KB_per_launch': f['FETCH_SIZE'], 'WRITE_SIZE_KB_per_launch': w['WRITE_SIZE'],
        'hbm_bytes_per_launch': hbm,
        'hbm_bytes_per_launch_rule': '(2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE reads 1/2 of a coalesced stream on gfx950 '
                                     '(MI355X_MICROARCH.md, HBM); WRITE_SIZE uncalibrated',
        'hbm_bytes_per_env_step': hbm / BATCH,
        'batch': BATCH,
        'sq_per_launch': q,
        'sq_shares_of_wave_cycles': {k: q[k] / q['SQ_WAVE_CYCLES'] for k in ('SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY')},
        'instructions_per_env_step': {k: q[k] / BATCH for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD')},
    }
    json.dump(out, open(os.path.join(dst, tag + '_pmc_summary.json'), 'w'), indent=1)
    for a, b in (('stats/bench_kernel_stats.csv', '_bench_rocprofv3_kernel_stats.csv'), ('bench.json', '_bench.json'),
                 ('bench_b32768.json', '_bench_b32768.json'), ('bench_no_launch_order.json', '_bench_no_launch_order.json'),
                 ('phase_profile_b4096.txt', '_phase_profile_b4096.txt'), ('phase_profile_b32768.txt', '_phase_profile_b32768.txt'),
                 ('stats_all/bench_all_kernel_stats.csv', '_bench_all_configs_rocprofv3_kernel_stats.csv'),
                 ('phase_profile_default14_b1024.txt', '_phase_profile_default14_b1024.txt'),
                 ('phase_profile_default14_b16384.txt', '_phase_profile_default14_b16384.txt'),
                 ('phase_profile_split_b1024.txt', '_phase_profile_split_b1024.txt'),
                 ('phase_profile_fdxb_b4096.txt', '_phase_profile_fdxb_b4096.txt'),
                 ('bench_2chronics.json', '_bench_2chronics.json'), ('bench_2chronics_b32768.json', '_bench_2chronics_b32768.json')):
        if os.path.exists(os.path.join(src, a)):
            shutil.copyfile(os.path.join(src, a), os.path.join(dst, tag + b))
    # the bench lines were printed before this summary existed: give them the traffic measured on the same build
    for b in ('_bench.json', '_bench_no_launch_order.json'):
        p = os.path.join(dst, tag + b)
        d = json.loads(open(p).read().strip().split('\n')[-1])
        d['roofline']['traffic'] = hbm
        open(p, 'w').write(json.dumps(d) + '\n')
    print('HBM bytes per launch %.0f (%.1f KB per env-step); wave cycles: %s' % (hbm, hbm / BATCH / 1024, out['sq_shares_of_wave_cycles']))


if __name__ == '__main__':
    main()

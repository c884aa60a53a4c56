#!/usr/bin/env python
"""Builds the data fixtures under tests/golden/ from the read-only reference tree.

Runs ONLY in the build container (needs /root/reference); the GPU box uses the committed outputs.
What is produced is data, not code:
  * envs/<name>/level0/reference_grid.json   bus/gen/branch arrays of the reference case (JSON lists)
  * envs/<name>/level0/configuration.yaml    the parsed scalar settings, re-emitted by yaml.safe_dump
  * envs/<name>/level0/chronics/<id>/*.csv   the reference tests' own 18-row chronic data files
  * envs/<name>/level0/chronics/<id>.npz     compact float32 cache of a shipped chronic (first rows)
"""
import os
import shutil
import sys

import yaml

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from pypownet_amd.case import load_case_file, save_case_json  # noqa: E402
from pypownet_amd.chronic import Chronic  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'envs')


def emit_env(src_env, dst_name, chronics, npz=False, max_rows=None, level='level0'):
    src = os.path.join(REF, src_env, level)
    dst = os.path.join(OUT, dst_name, level)
    os.makedirs(os.path.join(dst, 'chronics'), exist_ok=True)
    save_case_json(load_case_file(os.path.join(src, 'reference_grid.py')), os.path.join(dst, 'reference_grid.json'))
    with open(os.path.join(src, 'configuration.yaml')) as f:
        conf = yaml.safe_load(f)
    with open(os.path.join(dst, 'configuration.yaml'), 'w') as f:
        yaml.safe_dump(conf, f, default_flow_style=False, sort_keys=False)
    for c in chronics:
        csrc = os.path.join(src, 'chronics', c)
        if npz:
            ch = Chronic(csrc)
            rows = ch.n_timesteps + 1 if max_rows is None else max_rows  # +1: planned shift needs row t+1
            ch.to_npz(os.path.join(dst, 'chronics', c + '.npz'), max_rows=rows)
        else:
            cdst = os.path.join(dst, 'chronics', c)
            os.makedirs(cdst, exist_ok=True)
            for fn in sorted(os.listdir(csrc)):
                if fn.endswith('.csv'):
                    shutil.copyfile(os.path.join(csrc, fn), os.path.join(cdst, fn))


if __name__ == '__main__':
    for t in ['default14_for_tests', 'default14_for_tests_alpha', 'default14_for_tests_beta',
              'default14_for_tests_hard_overflow']:
        emit_env(os.path.join('tests', 'parameters', t), t, ['a'])
    # all twelve chronics of the benchmark environments (environment e plays chronic e mod 12)
    twelve = list('abcdefghijkl')
    emit_env(os.path.join('parameters', 'default14'), 'default14', twelve, npz=True)
    emit_env(os.path.join('parameters', 'default118'), 'default118', twelve, npz=True)
    emit_env(os.path.join('parameters', 'default30'), 'default30', ['a', 'b', 'c'], npz=True)
    print('fixtures written to', os.path.abspath(OUT))

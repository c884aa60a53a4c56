#!/usr/bin/env python
"""Reads the two rocprofv3 --pmc passes over build/hbm_counter_calib (WRITE_SIZE, FETCH_SIZE) and prints, per calibration
kernel, the counter value per launch next to the bytes the kernel is known to move -> JSON (kept in the PMC summary of the round).

    python tools/ubench/hbm_counter_calib.py <dir of the WRITE_SIZE pass> <dir of the FETCH_SIZE pass>"""
import collections
import csv
import glob
import json
import os
import sys

MIB = 1 << 20
KNOWN = {   # kernel -> (bytes written, bytes read) per launch
    'calib_write16': (256 * MIB, 0), 'calib_write8': (256 * MIB, 0), 'calib_write1': (64 * MIB, 0),
    'calib_write_row': ((64 * MIB // 186) * 186, 0), 'calib_read16': (0, 256 * MIB),
}


def per_launch(d, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != counter:
                continue
            for k in KNOWN:
                if k in r['Kernel_Name']:
                    acc[k] += float(r['Counter_Value'])
                    n[k].add(r['Dispatch_Id'])
    return {k: acc[k] / len(n[k]) for k in acc}


def main():
    w = per_launch(sys.argv[1], 'WRITE_SIZE')
    f = per_launch(sys.argv[2], 'FETCH_SIZE')
    out = {'note': 'counter value per launch (rocprofv3 reports KB) x 1024 / bytes the kernel is known to move; '
                   '1.0 = the counter counts every byte once, in KB'}
    for k, (bw, br) in KNOWN.items():
        e = {'bytes_written': bw, 'bytes_read': br, 'WRITE_SIZE_KB': w.get(k), 'FETCH_SIZE_KB': f.get(k)}
        if bw and w.get(k) is not None:
            e['WRITE_SIZE_x1024_per_byte_written'] = w[k] * 1024 / bw
        if br and f.get(k) is not None:
            e['FETCH_SIZE_x1024_per_byte_read'] = f[k] * 1024 / br
        out[k] = e
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""profiles/hbm_traffic.json from the separate rocprofv3 --pmc passes of tools/collect_profiles.sh:
python tools/make_hbm_traffic.py profiles/r03_pmc  (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU of the fbank-40 instantiation
of fbank512b_kernel; the MFCC-13 instantiation of the same run is reported beside it)"""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
FBANK, MFCC = 'fbank512b_kernel<13, 1,', 'fbank512b_kernel<13, 2,'   # <NJ, KIND (1 fbank, 2 mfcc), ..>
acc, acc_mfcc = collections.defaultdict(list), collections.defaultdict(list)
for name in sorted(glob.glob(os.path.join(src, 'pmc_group_*.csv'))):
    for r in csv.DictReader(open(name)):
        if FBANK in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
        if MFCC in r['Kernel_Name']:
            acc_mfcc[r['Counter_Name']].append(float(r['Counter_Value']))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
mean_mfcc = {k: sum(v) / len(v) for k, v in acc_mfcc.items()}
fetch_kib, write_kib = mean['FETCH_SIZE'], mean['WRITE_SIZE']
frames = 10000 * 298
out = {
    '_comment': (
        'HBM traffic of fbank512b_kernel<13, 1, 0, true> per launch (10 000 x 3 s utterances, fbank-40), '
        'from two separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) with --kernel-trace only '
        '(tools/collect_profiles.sh), corrected as MI355X_MICROARCH.md prescribes: counters are in KiB; on '
        'gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read (TCC_EA0_RDREQ x 64 B for '
        '128-B requests) so it is doubled; WRITE_SIZE matches the known output byte count 1:1 '
        '(2 980 000 x 160 B = 465 625 KiB) and is used as is.  Written by tools/make_hbm_traffic.py from '
        + src + '/pmc_group_1.csv, pmc_group_2.csv, pmc_group_3.csv.'),
    'workload': {'kind': 'fbank40', 'utterances': 10000, 'seconds': 3.0},
    'FETCH_SIZE_KiB': fetch_kib,
    'WRITE_SIZE_KiB': write_kib,
    'read_bytes': int(round(2 * fetch_kib * 1024)),
    'write_bytes': int(round(write_kib * 1024)),
    'traffic_bytes_per_launch': int(round(2 * fetch_kib * 1024 + write_kib * 1024)),
    'algorithmic_bytes_per_launch': frames * 480,
    'mfcc13': {
        '_comment': 'fbank512b_kernel<13, 2, 1, true> (MFCC-13 with raw energy) in the same runs: 372 B/frame algorithmic',
        'FETCH_SIZE_KiB': mean_mfcc.get('FETCH_SIZE'), 'WRITE_SIZE_KiB': mean_mfcc.get('WRITE_SIZE'),
        'traffic_bytes_per_launch': (int(round(2 * mean_mfcc['FETCH_SIZE'] * 1024 + mean_mfcc['WRITE_SIZE'] * 1024))
                                     if 'FETCH_SIZE' in mean_mfcc and 'WRITE_SIZE' in mean_mfcc else None),
        'algorithmic_bytes_per_launch': frames * 372,
        'valu_wave_instrs_per_launch': mean_mfcc.get('SQ_INSTS_VALU')},
    'valu': {
        '_comment': (
            'SQ_INSTS_VALU per launch of the same workload (separate --pmc pass) against the f32 VALU issue '
            'rate of the data sheet: one wave64 instruction per 2 clocks per SIMD at 2.4 GHz over 1024 SIMDs '
            '= 1228.8 G wave-instr/s (MI355X_MICROARCH.md; plain VOP2 streams measure 2.3-2.9 clocks at 4 '
            'waves per SIMD, 2.55 at 8; DPP, SDWA conversions, v_cndmask with an SGPR mask, v_med3 and '
            'v_dot2c issue at about half that rate)'),
        'wave_instrs_per_launch': mean['SQ_INSTS_VALU'],
        'peak_wave_instrs_per_s': 1228.8e9,
    },
}
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles',
                       'hbm_traffic.json'), 'w') as fh:
    json.dump(out, fh, indent=2)
    fh.write('\n')
print(json.dumps({k: v for k, v in out.items() if not k.startswith('_') and k != 'valu'}))

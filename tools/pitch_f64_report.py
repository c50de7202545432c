#!/usr/bin/env python
"""profiles/r03_pitch_f64.txt: the C pitch oracle against the float64 restatement (oracle/spec_f64.pitch) on
test.wav and seeded synthetic utterances.  CPU only:  python tools/pitch_f64_report.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from test_oracle_pins import pitch_oracle_vs_f64  # noqa: E402
from shennong_amd import synth  # noqa: E402
from shennong_amd.audio import Audio  # noqa: E402

wave = Audio.load(os.path.join(ROOT, 'tests', 'golden', 'test.wav')).data
rows = [('test.wav (140 frames)', pitch_oracle_vs_f64([wave])),
        ('50 synthetic 2 s utterances', pitch_oracle_vs_f64(
            [synth.utterances(900 + i, 1, 32000)[0] for i in range(50)])),
        ('20 synthetic 3.0-4.2 s utterances', pitch_oracle_vs_f64(
            [synth.utterances(300 + i, 1, 48000 + 997 * i)[0] for i in range(20)]))]
print('%-36s %-28s %-36s %s' % ('', 'max |resampled NCCF diff|', 'frames with another Viterbi state',
                                'largest distance'))
for name, (worst, differ, total, step) in rows:
    print('%-36s %-28.2e %-36s %d state(s)' % (name, worst, '%d of %d' % (differ, total), step))

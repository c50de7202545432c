"""Dev helper: latency of processor.process(audio) on ONE utterance (the reference's calling pattern,
shennong/processor/base.py:376-436), host to host"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shennong_amd import Audio, synth
from shennong_amd.processor import (FilterbankProcessor, MfccProcessor, PlpProcessor, SpectrogramProcessor,
                                    KaldiPitchProcessor, EnergyProcessor)
from shennong_amd.postprocessor import DeltaPostProcessor, CmvnPostProcessor
for seconds in (3, 30):
    wave = synth.utterances(1, 1, 16000 * seconds)[0]
    audio = Audio(wave, 16000)
    for proc in (FilterbankProcessor(num_bins=40), FilterbankProcessor(num_bins=40, dither=0), MfccProcessor(dither=0),
                 PlpProcessor(dither=0), SpectrogramProcessor(dither=0), EnergyProcessor(dither=0), KaldiPitchProcessor()):
        for _ in range(5):
            feats = proc.process(audio)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            feats = proc.process(audio)
            ts.append(time.perf_counter() - t0)
        print('%2d s  %-12s dither %-4s: process() median %.3f ms, min %.3f ms  (%d x %d)' % (
            seconds, proc.name, getattr(proc, 'dither', '-'), 1e3 * np.median(ts), 1e3 * np.min(ts), *feats.shape), flush=True)
    mf = MfccProcessor(dither=0).process(audio)
    d = DeltaPostProcessor()
    for _ in range(5):
        d.process(mf)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        d.process(mf)
        ts.append(time.perf_counter() - t0)
    print('%2d s  delta: process() median %.3f ms' % (seconds, 1e3 * np.median(ts)))

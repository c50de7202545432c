#!/usr/bin/env python
"""N independent single-rank processes of the streamed pipeline on ONE box: what eight
ranks of a node do to each other on the host side - page-locked memory, memory bandwidth, the interpreter's own
threads.  No RCCL, no exchange: each process owns total_hours / N of the BASELINE config 5 corpus (pinned index,
CMVN by speaker, VAD weights, fbank-40 + delta + pitch) and they start together.  On a one-GPU box the processes
also share the GPU, whose work is the same in sum for every N: the table shows whether the HOST scales.

    python tools/r06_multiproc_streamed.py [total_hours] > profiles/r06_streamed_multiprocess.txt
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(hours, njobs, tag, sync_dir, ark=None):
    sys.path.insert(0, ROOT)
    from shennong_amd import Audio, Utterances, pipeline, synth
    from shennong_amd.logger import get_logger
    waves = synth.utterances(0, 1000, 48000)
    n = int(hours * 1200)
    cfg = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
    cfg['filterbank']['num_bins'] = 40
    cfg['filterbank']['dither'] = 0
    cfg['cmvn']['by_speaker'] = True
    audios = [Audio(waves[i], 16000, validate=False) for i in range(len(waves))]
    index = Utterances([(f'u{i:06d}', audios[i % len(audios)], f's{i % 1000:04d}') for i in range(n)]).pin()
    quiet = get_logger('bench', 'error')
    seen = [0]

    def sink(feats):
        seen[0] += len(feats)
    warm = Utterances([(f'w{i:06d}', audios[i % len(audios)], f's{i % 1000:04d}') for i in range(min(n, 9600))]).pin()
    pipeline.extract_features_streamed(cfg, warm, lambda f: None, log=quiet, njobs=njobs)
    del warm
    open(os.path.join(sync_dir, 'ready%s' % tag), 'w').close()
    while not os.path.exists(os.path.join(sync_dir, 'go')):
        time.sleep(0.001)
    writer = None
    if ark:   # every process writes its own shard (float matrices): separate files, separate locks
        from shennong_amd.serializers import KaldiStreamWriter
        writer = KaldiStreamWriter(os.path.join(ark, 'shard%s.ark' % tag), double=False)

        def sink(feats, write=writer.write):   # noqa: F811
            seen[0] += len(feats)
            write(feats)
    t0 = time.perf_counter()
    pipeline.extract_features_streamed(cfg, index, sink, log=quiet, njobs=njobs)
    if writer is not None:
        writer.close()
    dt = time.perf_counter() - t0
    assert seen[0] == n
    print(json.dumps({'tag': tag, 'hours': hours, 'wall_s': dt}), flush=True)


def main():
    total = float(sys.argv[1]) if len(sys.argv) > 1 else 125.0
    ark = sys.argv[2] if len(sys.argv) > 2 else ''
    print('# processes x njobs: aggregate hours of audio per second = total hours / slowest process '
          '(%.0f h in all, split evenly; pinned indexes; nproc %d)%s' % (
              total, os.cpu_count(), '; every process writes its shard to a Kaldi archive of float matrices under '
              + ark if ark else ''))
    for njobs in ((2,) if ark else (1, 2)):
        for procs in (1, 2, 4, 8):
            with tempfile.TemporaryDirectory() as sync_dir, tempfile.TemporaryDirectory(dir=ark or None) as out:
                kids = [subprocess.Popen([sys.executable, __file__, 'child', str(total / procs), str(njobs), str(k),
                                          sync_dir, out if ark else ''], stdout=subprocess.PIPE, text=True)
                        for k in range(procs)]
                while sum(os.path.exists(os.path.join(sync_dir, 'ready%d' % k)) for k in range(procs)) < procs:
                    if any(kid.poll() not in (None, 0) for kid in kids):
                        raise SystemExit('a child failed')
                    time.sleep(0.01)
                open(os.path.join(sync_dir, 'go'), 'w').close()
                walls = [json.loads(kid.communicate()[0].strip().splitlines()[-1])['wall_s'] for kid in kids]
            print('processes %d  njobs %d: slowest %.3f s, fastest %.3f s -> %6.1f h/s aggregate' % (
                procs, njobs, max(walls), min(walls), total / max(walls)), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(float(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6] if len(sys.argv) > 6 else None)
    else:
        main()

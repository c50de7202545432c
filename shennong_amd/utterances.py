"""Utterances index: ``<id> <audio> [<speaker>] [<tstart> <tstop>]``

Mirror of reference shennong/utterances.py:37-346 (formats, validation, duration bookkeeping,
grouping by speaker, index files, `fit_to_duration`); `audio` may be a wav path or an in-memory
:class:`Audio` (the benchmark feeds arrays directly).
"""

import collections
import os
import random
import warnings

from shennong_amd.audio import Audio

VALID_FORMATS = {
    1: '<utterance-id> <audio-file>',
    2: '<utterance-id> <audio-file> <speaker-id>',
    3: '<utterance-id> <audio-file> <tstart> <tstop>',
    4: '<utterance-id> <audio-file> <speaker-id> <tstart> <tstop>'}


# optional fields after ``<id> <audio>``, by number of fields
_LAYOUTS = {2: (), 3: ('speaker',), 4: ('tstart', 'tstop'), 5: ('speaker', 'tstart', 'tstop')}


def _seconds(value, what):
    try:
        return None if value is None else float(value)
    except ValueError:
        raise ValueError(f'cannot cast {what} as float: {value}') from None


class Utterance:
    """One line of the index: a name, an audio file (or an in-memory :class:`Audio`), optionally a
    speaker and a (tstart, tstop) interval of the file in seconds"""
    def __init__(self, *fields):
        if len(fields) not in _LAYOUTS:
            raise ValueError(f'invalid utterance format: {fields}')
        self._format = len(fields) - 1
        self._name, self._audio = fields[:2]
        optional = dict(zip(_LAYOUTS[len(fields)], fields[2:]))
        self._speaker = optional.get('speaker')
        if ('tstart' in optional) and (optional['tstart'] is None) != (optional['tstop'] is None):
            raise ValueError('both tstart and tstop must be defined or None')
        self._tstart = _seconds(optional.get('tstart'), 'tstart')
        self._tstop = _seconds(optional.get('tstop'), 'tstop')
        if self._tstart is not None and not 0 <= self._tstart < self._tstop:
            raise ValueError(
                'we must have 0 <= tstart < tstop, but '
                f'(tstart, tstop)=({self._tstart}, {self._tstop})')
        # the duration comes from the file itself (scanning raises if it is missing or not audio);
        # an interval that runs past the end of the file is cut there
        self._scan = Audio.scan(self._audio)   # (kept: the pipeline sizes its batches from it without a second scan)
        self._duration = self._scan.duration
        if self._tstart is not None:
            if self._tstop > self._duration:
                warnings.warn(
                    f'{self._audio}: file duration is {self._duration} but '
                    f'asking interval ({self._tstart}, {self._tstop}), '
                    f'will be truncated')
                self._tstop = self._duration
            self._duration = self._tstop - self._tstart

    name = property(lambda self: self._name)
    audio_file = property(lambda self: self._audio)
    speaker = property(lambda self: self._speaker)
    tstart = property(lambda self: self._tstart)
    tstop = property(lambda self: self._tstop)
    format = property(lambda self: self._format)
    duration = property(lambda self: self._duration)

    def __str__(self):
        """The index line of this utterance"""
        fields = [self.name, self.audio_file]
        fields += [getattr(self, field) for field in _LAYOUTS[self._format + 1]]
        return ' '.join(str(field) for field in fields)

    def __eq__(self, other):
        return str(self) == str(other)

    def __hash__(self):
        return hash(str(self))

    def load_audio(self):
        """The audio of the utterance: the file, or its (tstart, tstop) interval"""
        audio = self._audio if isinstance(self._audio, Audio) else Audio.load(self._audio)
        if self.tstart or self.tstop:
            audio = audio.segment([(self.tstart, self.tstop)])[0]
        return audio


class Utterances:
    def __init__(self, utterances):
        parsed = []
        for utt in utterances:
            if not isinstance(utt, Utterance):
                try:
                    utt = Utterance(*utt)
                except TypeError:
                    raise ValueError(
                        f'utterance must be an iterable, not {utt}') from None
            parsed.append(utt)
        if not parsed:
            raise ValueError('empty input utterances')
        if len(set(u.format for u in parsed)) != 1:
            raise ValueError('utterances format is not homogeneous')
        duplicates = [u for u, c in collections.Counter(
            u.name for u in parsed).items() if c > 1]
        if duplicates:
            raise ValueError(
                f'duplicates found in utterances: {", ".join(duplicates)}')
        self._format = parsed[0].format
        # sorted by audio file like the reference (its Audio.load cache); in-memory audio (this
        # backend's extension) keeps the insertion order
        if all(isinstance(u.audio_file, str) for u in parsed):
            parsed = sorted(parsed, key=lambda u: (u.audio_file, u.name))
        self._utterances = {u.name: u for u in parsed}

    def __len__(self):
        return len(self._utterances)

    def __iter__(self):
        return iter(self._utterances.values())

    def __getitem__(self, name):
        return self._utterances[name]

    def __eq__(self, other):
        return list(self) == list(other)

    @classmethod
    def load(cls, filename):
        """Utterances from an index file, one ``<id> <audio> [<speaker>] [<tstart> <tstop>]`` per
        line, all lines in the same format"""
        if not os.path.isfile(filename):
            raise ValueError(f'{filename} not found')
        with open(filename, 'r') as stream:
            lines = (line.strip() for line in stream.readlines())
        return cls([line.split(' ') for line in lines if line])

    def save(self, filename):
        """Writes the utterances index to `filename`"""
        with open(filename, 'w') as stream:
            stream.write('\n'.join(str(utt) for utt in self) + '\n')

    def fit_to_duration(self, duration, truncate=False, shuffle=False):
        """A subset of the utterances keeping `duration` seconds per speaker (reference
        utterances.py:348-418); raises ValueError without speakers, for a non-positive duration, or
        when a speaker has not enough audio and `truncate` is False (a warning when it is True)"""
        if duration <= 0:
            raise ValueError(
                f'duration must be a positive number, it is {duration}')
        kept = []
        for speaker, utterances in self.by_speaker().items():
            if shuffle:
                random.shuffle(utterances)
            missing = duration  # seconds still to collect for this speaker
            for utt in utterances:
                if missing <= 0:
                    break
                tstart = utt.tstart or 0
                take = min(utt.duration, missing)
                kept.append(Utterance(utt.name, utt.audio_file, utt.speaker, tstart, tstart + take))
                missing -= take
            if missing > 0:
                message = (
                    f'speaker {speaker}: only {duration - missing}s'
                    f' of audio available but {duration}s requested')
                if not truncate:
                    raise ValueError(message)
                warnings.warn(message)
        return Utterances(kept)

    def format(self, type=int):
        """The utterances format: its code (`type` int) or its description (`type` str)"""
        return VALID_FORMATS[self._format] if type is str else self._format

    def has_speakers(self):
        """True if there is speaker information"""
        return self.format(type=int) in (2, 4)

    def by_speaker(self):
        """Dictionary speaker -> list of :class:`Utterance`"""
        if not self.has_speakers():
            raise ValueError('utterances have no speaker information')
        by_speaker = collections.defaultdict(list)
        for utt in self:
            by_speaker[utt.speaker].append(utt)
        return by_speaker

    def by_name(self):
        return self._utterances

    def pin(self):
        """The same utterances with their audio loaded ONCE, as 16-bit integers - what every processor forces
        a signal to before it runs (reference processor/base.py:428) -, into one page-locked block
        (``_backend.PinnedCorpus``); segments are cut, files are not read again.  ``process_all`` on the
        result uploads straight from that block: no gather into a staging buffer and no per-utterance
        checks (they are made here: mono, one sample rate).  An extension of this backend - the reference
        has no device to feed.  Needs the HIP library (page-locked memory comes from it)."""
        import numpy as np
        from shennong_amd import _backend
        from shennong_amd.audio import load_int16_block, sample_range
        utts = list(self)
        if all(isinstance(u.audio_file, str) for u in utts):
            # WAV files: header scans, then the samples of all of them side by side straight into the block
            # (16-bit mono PCM natively, snf_wav_read_pcm16; other sample types through Audio.load + astype)
            scans = {}
            metas = [scans.get(u.audio_file) or scans.setdefault(
                u.audio_file, getattr(u, '_scan', None) or Audio.scan(u.audio_file)) for u in utts]
            rates = sorted(set(m.sample_rate for m in metas))
            if len(rates) != 1:
                raise ValueError('utterances to pin must share one sample rate, found ' +
                                 ', '.join('%dHz' % r for r in rates))
            for utt, meta in zip(utts, metas):
                if meta.nchannels != 1:
                    raise ValueError('signal must have one dimension, but it has {} ({})'.format(
                        meta.nchannels, utt.name))
            lengths = [sample_range(m.nsamples, m.sample_rate, u.tstart, u.tstop)[1] for u, m in zip(utts, metas)]
            corpus = _backend.PinnedCorpus(None, rates[0], lengths=lengths,
                                           fill=lambda block, soff: load_int16_block(utts, metas, block, soff))
        else:
            signals = [u.load_audio() for u in utts]
            rates = sorted(set(s.sample_rate for s in signals))
            if len(rates) != 1:
                raise ValueError('utterances to pin must share one sample rate, found ' +
                                 ', '.join('%dHz' % r for r in rates))
            for utt, signal in zip(utts, signals):
                if signal.nchannels != 1:
                    raise ValueError('signal must have one dimension, but it has {} ({})'.format(
                        signal.nchannels, utt.name))
            corpus = _backend.PinnedCorpus([s.astype(np.int16).data for s in signals], rates[0])
        fields = (lambda u: (u.speaker,)) if self.has_speakers() else (lambda u: ())
        pinned = Utterances([(u.name, Audio(view, rates[0], validate=False)) + fields(u)
                             for u, view in zip(utts, corpus.views)])
        pinned._pinned = corpus
        return pinned

    def duration(self):
        """Total duration of the utterances in seconds"""
        return sum(utt.duration for utt in self)

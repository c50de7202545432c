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


class Utterance:
    def __init__(self, *args):
        if len(args) < 2 or len(args) > 5:
            raise ValueError(f'invalid utterance format: {args}')
        self._format = len(args) - 1
        self._name, self._audio = args[0], args[1]
        self._speaker = self._tstart = self._tstop = None
        if len(args) == 3:
            self._speaker = args[2]
        elif len(args) == 4:
            self._tstart, self._tstop = args[2], args[3]
        elif len(args) == 5:
            self._speaker = args[2]
            self._tstart, self._tstop = args[3], args[4]

        if self._tstart is not None:
            try:
                self._tstart = float(self._tstart)
            except ValueError:
                raise ValueError(
                    f'cannot cast tstart as float: {self._tstart}') from None
        if self._tstop is not None:
            try:
                self._tstop = float(self._tstop)
            except ValueError:
                raise ValueError(
                    f'cannot cast tstop as float: {self._tstop}') from None
        if (self._tstart is None) != (self._tstop is None):
            raise ValueError('both tstart and tstop must be defined or None')
        if self._tstart is not None and (
                self._tstart < 0 or self._tstart >= self._tstop):
            raise ValueError(
                'we must have 0 <= tstart < tstop, but '
                f'(tstart, tstop)=({self._tstart}, {self._tstop})')

        # utterance duration; scanning raises if the file is not found nor valid
        self._duration = Audio.scan(self._audio).duration
        if self._tstart is not None:
            if self._tstop > self._duration:
                warnings.warn(
                    f'{self._audio}: file duration is {self._duration} but '
                    f'asking interval ({self._tstart}, {self._tstop}), '
                    f'will be truncated')
                self._tstop = self._duration
            self._duration = self._tstop - self._tstart

    def __eq__(self, other):
        return str(self) == str(other)

    def __hash__(self):
        return hash(str(self))

    name = property(lambda self: self._name)
    audio_file = property(lambda self: self._audio)
    speaker = property(lambda self: self._speaker)
    tstart = property(lambda self: self._tstart)
    tstop = property(lambda self: self._tstop)
    format = property(lambda self: self._format)
    duration = property(lambda self: self._duration)

    def __str__(self):
        if self._format == 1:
            return f'{self.name} {self.audio_file}'
        if self._format == 2:
            return f'{self.name} {self.audio_file} {self.speaker}'
        if self._format == 3:
            return f'{self.name} {self.audio_file} {self.tstart} {self.tstop}'
        return (f'{self.name} {self.audio_file} {self.speaker} '
                f'{self.tstart} {self.tstop}')

    def load_audio(self):
        data = (self._audio if isinstance(self._audio, Audio)
                else Audio.load(self._audio))
        if self.tstart or self.tstop:
            data = data.segment([(self.tstart, self.tstop)])[0]
        return data


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
        segments = []
        for speaker, utterances in self.by_speaker().items():
            if shuffle:
                random.shuffle(utterances)
            remaining_duration = duration
            for utt in utterances:
                tstart = 0 if utt.tstart is None else utt.tstart
                tstop = utt.duration - tstart if utt.tstop is None else utt.tstop
                if utt.duration >= remaining_duration:
                    segments.append(Utterance(
                        utt.name, utt.audio_file, utt.speaker, tstart,
                        tstart + remaining_duration))
                    remaining_duration = 0
                    break
                segments.append(Utterance(
                    utt.name, utt.audio_file, utt.speaker, tstart, tstop))
                remaining_duration -= utt.duration
            if remaining_duration > 0:
                message = (
                    f'speaker {speaker}: only {duration - remaining_duration}s'
                    f' of audio available but {duration}s requested')
                if truncate:
                    warnings.warn(message)
                else:
                    raise ValueError(message)
        return Utterances(segments)

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

    def duration(self):
        """Total duration of the utterances in seconds"""
        return sum(utt.duration for utt in self)

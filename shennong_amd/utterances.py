"""Minimal utterances index: ``<id> <audio> [<speaker>] [<tstart> <tstop>]``

Enough of reference shennong/utterances.py:37-260 to drive ``process_all``; `audio` may be a wav
path or an in-memory :class:`Audio` (the benchmark feeds arrays directly).
"""

import collections

from shennong_amd.audio import Audio


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
            self._tstart, self._tstop = float(args[2]), float(args[3])
        elif len(args) == 5:
            self._speaker = args[2]
            self._tstart, self._tstop = float(args[3]), float(args[4])
        if self._tstart is not None and (
                self._tstart < 0 or self._tstart >= self._tstop):
            raise ValueError(
                'we must have 0 <= tstart < tstop, but '
                f'(tstart, tstop)=({self._tstart}, {self._tstop})')

    name = property(lambda self: self._name)
    audio_file = property(lambda self: self._audio)
    speaker = property(lambda self: self._speaker)
    tstart = property(lambda self: self._tstart)
    tstop = property(lambda self: self._tstop)
    format = property(lambda self: self._format)

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
        self._utterances = {u.name: u for u in parsed}

    def __len__(self):
        return len(self._utterances)

    def __iter__(self):
        return iter(self._utterances.values())

    def __getitem__(self, name):
        return self._utterances[name]

    def by_name(self):
        return dict(self._utterances)

"""Declarative processor parameters

Every processor of this package owns ONE C option record (``snf_options``, shennong_amd/_abi.py)
that is handed to the library by value.  A public parameter of a processor is a view on one field
of that record: instead of a hand-written getter / setter pair per parameter, the classes declare

    dither = Option('frame.dither', 'Amount of dithering, 0.0 means no dither', F32)

and :class:`Option` (a descriptor) does the reading, writing and type conversion.  The conversions
reproduce what a user of the reference observes on its Kaldi-backed attributes (floats come back as
``np.float32`` because they went through a C float, flags as ``bool``, durations are given in seconds
and stored in milliseconds).
"""

import inspect

import numpy as np

from shennong_amd import _abi

RAW, F32, FLAG, SECONDS_F32, SECONDS = 'raw', 'f32', 'flag', 'seconds_f32', 'seconds'

_READ = {
    RAW: lambda v: v,
    F32: np.float32,
    FLAG: bool,
    SECONDS_F32: lambda v: np.float32(v / 1000.0),
    SECONDS: lambda v: v / 1000.0,
}
_WRITE = {
    RAW: lambda v: v,
    F32: lambda v: v,
    FLAG: bool,
    SECONDS_F32: lambda v: v * 1000.0,
    SECONDS: lambda v: v * 1000.0,
}


class Option:
    """A processor parameter stored in field `path` (dotted, e.g. ``'frame.dither'``) of the
    processor's option record ``self._record``"""
    def __init__(self, path, doc, kind=RAW, check=None):
        *self._parents, self._field = path.split('.')
        self._read, self._write = _READ[kind], _WRITE[kind]
        self._check = check  # check(value): raises ValueError for a value the option cannot take
        self.__doc__ = doc

    def _holder(self, obj):
        holder = obj._record
        for name in self._parents:
            holder = getattr(holder, name)
        return holder

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        return self._read(getattr(self._holder(obj), self._field))

    def __set__(self, obj, value):
        if self._check is not None:
            self._check(value)
        setattr(self._holder(obj), self._field, self._write(value))


def require(condition, message):
    """A `check` for :class:`Option`: ``require(lambda v: v >= 0, 'x must be >= 0, it is {}')``"""
    def check(value):
        if not condition(value):
            raise ValueError(message.format(value))
    return check


class Configurable:
    """Mixin for the processors: creates the option record of plan kind ``_kind`` and assigns the
    constructor arguments in the order of the signature"""
    _kind = None

    def _configure(self, arguments):
        """`arguments`: the ``locals()`` of the calling ``__init__``"""
        self._record = _abi.default_options(self._kind)
        for name in inspect.signature(type(self).__init__).parameters:
            if name != 'self':
                setattr(self, name, arguments[name])

    def _build_options(self):
        """The option record by value: later edits of the attributes do not reach a call that is
        under way, and every call sees the attributes as they are now (reference
        processor/base.py:421-425 re-forwards its option structs the same way)"""
        return _abi.copy_options(self._record)

"""Base class of all processors: sklearn-like get_params / set_params

Mirrors reference shennong/base.py:59-150 (parameters are the explicit keyword arguments of
``__init__``; nested ``a__b`` keys are forwarded to sub-processors).
"""

import abc
import collections
import inspect

from shennong_amd.logger import get_logger


class BaseProcessor:
    """Base class for all processors"""
    def __init__(self):
        self._logger = get_logger(self.name, level='info')

    def __repr__(self):
        return self.__class__.__name__

    @abc.abstractproperty
    def name(self):
        """Processor name"""

    @property
    def log(self):
        """Processor logger"""
        if not hasattr(self, '_logger'):
            self._logger = get_logger(self.name, level='info')
        return self._logger

    def set_logger(self, level,
                   formatter='%(levelname)s - %(name)s - %(message)s'):
        """Change level and/or format of the processor's logger"""
        self._logger = get_logger(self.name, level=level, formatter=formatter)

    @classmethod
    def _get_param_names(cls):
        cached = cls.__dict__.get('_param_names_cache')
        if cached is not None:
            return cached
        names = cls._inspect_param_names()
        cls._param_names_cache = names
        return names

    @classmethod
    def _inspect_param_names(cls):
        init = getattr(cls.__init__, 'deprecated_original', cls.__init__)
        if init is object.__init__:  # pragma: nocover
            return []
        signature = inspect.signature(init)
        parameters = [p for p in signature.parameters.values()
                      if p.name != 'self' and p.kind != p.VAR_KEYWORD]
        for param in parameters:
            if param.kind == param.VAR_POSITIONAL:
                raise RuntimeError(
                    f'processors should always specify their parameters in '
                    f'the signature of their __init__ (no varargs). {cls} '
                    f'with constructor {signature} does not follow this '
                    f'convention.')
        return sorted([p.name for p in parameters])

    def get_params(self, deep=True):
        """Get parameters for this processor as a dict name -> value"""
        out = dict()
        for key in self._get_param_names():
            value = getattr(self, key, None)
            if deep and hasattr(value, 'get_params'):
                out.update((key + '__' + k, val)
                           for k, val in value.get_params().items())
            out[key] = value
        return out

    def set_params(self, **params):
        """Set the parameters of this processor, returns self"""
        if not params:
            return self
        valid_params = self.get_params(deep=True)
        nested_params = collections.defaultdict(dict)
        for key, value in params.items():
            key, delim, sub_key = key.partition('__')
            if key not in valid_params:
                raise ValueError(
                    f'invalid parameter {key} for processor {self}, '
                    f'check the list of available parameters '
                    f'with `processor.get_params().keys()`.')
            if delim:
                nested_params[key][sub_key] = value
            else:
                try:
                    setattr(self, key, value)
                except AttributeError:
                    raise ValueError(f'cannot set attribute {key} for {self}')
                valid_params[key] = value
        for key, sub_params in nested_params.items():
            valid_params[key].set_params(**sub_params)
        return self

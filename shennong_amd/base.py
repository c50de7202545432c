"""Base class of all processors: parameters as constructor keywords

Every processor declares its parameters as explicit keyword arguments of ``__init__`` and exposes
them as attributes of the same names; ``get_params`` / ``set_params`` (the scikit-learn estimator
protocol, as in reference shennong/base.py:59-150) read and write them by name, ``a__b`` addressing
parameter ``b`` of the processor stored in parameter ``a``.
"""

import inspect

from shennong_amd.logger import get_logger

_FORMAT = '%(levelname)s - %(name)s - %(message)s'


class BaseProcessor:
    """Base class for all processors"""
    name = None  # every processor class names itself

    def __init__(self):
        self._logger = get_logger(self.name, level='info')

    def __repr__(self):
        return type(self).__name__

    @property
    def log(self):
        """Processor logger (created on first use when a subclass did not call ``__init__``)"""
        if '_logger' not in self.__dict__:
            self._logger = get_logger(self.name, level='info')
        return self._logger

    def set_logger(self, level, formatter=_FORMAT):
        """Change level and/or format of the processor's logger"""
        self._logger = get_logger(self.name, level=level, formatter=formatter)

    # ---- parameters -------------------------------------------------------------------------------
    @classmethod
    def _get_param_names(cls):
        """Sorted names of the constructor's keyword parameters (cached per class: the pipeline asks
        once per utterance)"""
        names = cls.__dict__.get('_param_names')
        if names is None:
            names = []
            for param in inspect.signature(cls.__init__).parameters.values():
                if param.kind is param.VAR_POSITIONAL:
                    raise RuntimeError(
                        f'processors should always specify their parameters in the signature of '
                        f'their __init__ (no varargs): {cls} does not')
                if param.name != 'self' and param.kind is not param.VAR_KEYWORD:
                    names.append(param.name)
            names = cls._param_names = sorted(names)
        return names

    def get_params(self, deep=True):
        """Parameters of this processor as a dict name -> value; with `deep`, the parameters of
        processor-valued parameters appear too, as ``name__subname``"""
        params = {}
        for name in self._get_param_names():
            value = params[name] = getattr(self, name, None)
            if deep and hasattr(value, 'get_params'):
                for sub_name, sub_value in value.get_params().items():
                    params[f'{name}__{sub_name}'] = sub_value
        return params

    def set_params(self, **params):
        """Set parameters by name (``name__subname`` reaches into a processor-valued parameter);
        returns self.  Raises ValueError for a name that is not a parameter."""
        known = self.get_params(deep=False) if params else {}
        nested = {}
        for key, value in params.items():
            name, _, sub_name = key.partition('__')
            if name not in known:
                raise ValueError(
                    f'invalid parameter {name} for processor {self}, check the list of available '
                    f'parameters with `processor.get_params().keys()`.')
            if sub_name:
                nested.setdefault(name, {})[sub_name] = value
                continue
            try:
                setattr(self, name, value)
            except AttributeError:
                raise ValueError(f'cannot set attribute {name} for {self}') from None
        for name, sub_params in nested.items():
            getattr(self, name).set_params(**sub_params)
        return self

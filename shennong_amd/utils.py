"""Small host utilities (mirrors reference shennong/utils.py:18-96)"""

import copy
import multiprocessing
import threading

import numpy as np

from shennong_amd.logger import null_logger


def get_njobs(njobs=None, log=null_logger()):
    """Returns min(njobs, ncpus); ``ValueError`` if `njobs` <= 0 (reference utils.py:18-55)"""
    max_njobs = multiprocessing.cpu_count()
    if njobs is None:
        return max_njobs
    if njobs <= 0:
        raise ValueError(
            'njobs must be strictly positive, it is {}'.format(njobs))
    if njobs > max_njobs:
        log.warning(
            'asking %d CPU cores but reducing to %d (max available)',
            njobs, max_njobs)
        return max_njobs
    return njobs


def array2list(seq):
    """Converts numpy arrays in `seq` into lists (reference utils.py:65-73, which stops at lists: an
    array inside a list of dicts made `dict_equal` raise there; lists are walked as well here)"""
    if isinstance(seq, dict):
        return {k: array2list(v) for k, v in seq.items()}
    if isinstance(seq, (list, tuple)):
        return [array2list(v) for v in seq]
    if isinstance(seq, np.ndarray):
        return seq.tolist()
    return seq


def copy_properties(value):
    """Independent copy of a properties tree (dicts / lists of strings, numbers and arrays): what
    ``copy.deepcopy`` returns for these, several times faster - with one Features per utterance the
    generic deepcopy dominated the host time of a batched launch"""
    kind = type(value)
    if kind is dict:
        return {k: copy_properties(v) for k, v in value.items()}
    if kind is list:
        return [copy_properties(v) for v in value]
    if kind is np.ndarray:
        return value.copy()
    if kind in (str, int, float, bool, type(None)) or isinstance(value, np.generic):
        return value
    return copy.deepcopy(value)


def dict_equal(dict1, dict2):
    """True if the two dicts (which may hold numpy arrays) are equal"""
    return array2list(dict1) == array2list(dict2)


class paused_gc:
    """Cyclic garbage collection paused while a batched call makes its thousands of small objects (one
    Features per utterance, views, tuples): none of them is garbage, and with a large corpus index alive every
    generation-2 pass of the collector walks hundreds of thousands of objects - measured as 5-8 ms of a 36 ms
    `process_all` inside bench.py (150 000-utterance index alive) against none in a fresh process.  Reference
    counting frees everything as before.

    The collector is process-wide and the batches of a streamed corpus enter this from several threads at once:
    a nesting count (under a lock) disables it with the first section to start and re-enables it - if it was
    enabled then - with the LAST one to end, not the first.  While any section is open no thread of the process
    collects cycles."""
    _lock = threading.Lock()
    _depth = 0
    _was = False

    def __enter__(self):
        import gc
        cls = paused_gc
        with cls._lock:
            if cls._depth == 0:
                cls._was = gc.isenabled()
                gc.disable()
            cls._depth += 1
        return self

    @staticmethod
    def collect_young():
        """One pass over the two young generations, when the collector was enabled before the pause: for a
        long-running section that has just dropped what it made"""
        if paused_gc._was:
            import gc
            gc.collect(1)

    def __exit__(self, *exc):
        cls = paused_gc
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0 and cls._was:
                import gc
                gc.enable()
        return False

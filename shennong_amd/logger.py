"""Per-processor stderr logger (host utility; mirrors reference shennong/logger.py:30-84)"""

import logging
import sys


def get_logger(name, level='info',
               formatter='%(levelname)s - %(name)s - %(message)s'):
    """Returns a logger writing to stderr at the given `level`"""
    levels = {
        'debug': logging.DEBUG, 'info': logging.INFO,
        'warning': logging.WARNING, 'error': logging.ERROR}
    try:
        level = levels[level]
    except KeyError:
        raise ValueError(
            'invalid logging level "{}", must be in {}'.format(
                level, ', '.join(levels.keys()))) from None
    log = logging.getLogger(name)
    log.handlers.clear()
    log.propagate = False
    handler = logging.StreamHandler(sys.stderr)
    handler.setFormatter(logging.Formatter(formatter))
    log.addHandler(handler)
    log.setLevel(level)
    return log


def null_logger():
    """A logger that discards every message"""
    log = logging.getLogger('null')
    log.handlers.clear()
    log.addHandler(logging.NullHandler())
    log.propagate = False
    return log

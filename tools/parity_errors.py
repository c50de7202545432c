#!/usr/bin/env python
"""Summarises the file written by tests/conftest.py::assert_close under SNF_PARITY_LOG: worst measured
error of every parity test against the oracle, next to the tolerance the test asserts."""
import collections
import json
import sys

rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    r = json.loads(line)
    key = r['test'].split('::')[-1].split('[')[0]
    cur = rows.setdefault(key, dict(n=0, max_abs=0.0, max_rel=0.0, need=0.0, rtol=r['rtol'], atol=r['atol']))
    cur['n'] += 1
    cur['max_abs'] = max(cur['max_abs'], r['max_abs'])
    cur['max_rel'] = max(cur['max_rel'], r['max_rel'])
    cur['need'] = max(cur['need'], r['needed_atol_at_rtol'])
    cur['atol'] = max(cur['atol'], r['atol'])
fam = collections.OrderedDict()
for line in open(sys.argv[1]):
    r = json.loads(line)
    cur = fam.setdefault(r.get('family', 'other'), dict(n=0, max_abs=0.0, need=0.0, rtol=0.0, atol=0.0,
                                                        inside=0, size=0))
    cur['n'] += 1
    cur['inside'] += r.get('inside_1e-4_rel', 0)
    cur['size'] += r['size'] if 'inside_1e-4_rel' in r else 0
    cur['max_abs'] = max(cur['max_abs'], r['max_abs'])
    cur['need'] = max(cur['need'], r['needed_atol_at_rtol'])
    cur['rtol'] = max(cur['rtol'], r['rtol'])
    cur['atol'] = max(cur['atol'], r['atol'])
print('per feature family (atol needed = largest excess over rtol |want|; tests/conftest.py asserts twice that):')
print('(inside = fraction of all compared elements within the north_star\'s flat 1e-4 relative tolerance, no absolute term)')
print('%-40s %6s %10s %12s %8s %8s %12s' % ('family', 'calls', 'max abs', 'atol needed', 'rtol', 'atol', 'inside'))
for key, c in fam.items():
    print('%-40s %6d %10.2e %12.2e %8.0e %8.1e %12.6f' % (key, c['n'], c['max_abs'], c['need'], c['rtol'], c['atol'],
                                                          c['inside'] / c['size'] if c['size'] else float('nan')))
print()
print('%-40s %6s %10s %10s %12s %8s %8s' % ('test', 'calls', 'max abs', 'max rel', 'atol needed', 'rtol', 'atol'))
for key, c in rows.items():
    print('%-40s %6d %10.2e %10.2e %12.2e %8.0e %8.0e' % (key, c['n'], c['max_abs'], c['max_rel'], c['need'], c['rtol'], c['atol']))

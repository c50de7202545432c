#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, %) of a rocprofv3 rocpd .db, like `--stats`.
Usage: python tools/rocprof_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f'# {path}')
        print(f'{"kernel":<100} {"calls":>6} {"total_us":>12} {"avg_us":>12} {"min_us":>12} {"max_us":>12}')
        rows = cur.execute(
            'select name, count(*), sum(end - start), avg(end - start), '
            'min(end - start), max(end - start) from kernels group by name '
            'order by 3 desc').fetchall()
        for name, calls, tot, avg, mn, mx in rows:
            print(f'{name[:100]:<100} {calls:>6} {tot / 1e3:>12.1f} {avg / 1e3:>12.2f} '
                  f'{mn / 1e3:>12.2f} {mx / 1e3:>12.2f}')
        try:
            pmc = cur.execute(
                'select k.name, p.counter_name, avg(p.value), count(*) from pmc_events p '
                'join kernels k on k.dispatch_id = p.dispatch_id group by 1, 2').fetchall()
            for r in pmc:
                print('PMC', r)
        except Exception as err:  # noqa
            pass


if __name__ == '__main__':
    main()

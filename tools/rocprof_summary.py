#!/usr/bin/env python3
"""Summarise a rocprofv3 (--kernel-trace --stats) rocpd SQLite result into a text table for profiles/."""
import re
import sqlite3
import sys


def main(db, out=None, steps=None):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(top_kernels)')]
    rows = [dict(zip(cols, r)) for r in cur.execute('select * from top_kernels')]
    tot = sum(r['total_duration'] for r in rows)
    lines = ['# rocprofv3 --kernel-trace --stats summary of %s' % db, '# total kernel time %.3f ms%s' % (tot / 1e3, (' over %s bench steps' % steps) if steps else ''),
             '%-92s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct')]
    for r in rows[:40]:
        name = re.sub(r'\(anonymous namespace\)::', '', r['name'])
        name = re.sub(r'\(.*', '', name)[:92]
        lines.append('%-92s %8d %12.1f %10.2f %6.2f%%' % (name, r['total_calls'], r['total_duration'], r['average'], r['percentage']))
    txt = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main(*sys.argv[1:])

#!/usr/bin/env python3
"""Print per-kernel mean PMC counter values from a rocprofv3 --pmc rocpd SQLite db."""
import sqlite3, sys, re, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = 'counters_collection' if 'counters_collection' in tabs else None
print([t for t in tabs if 'counter' in t.lower() or 'pmc' in t.lower()])
if view:
    cols = [r[1] for r in cur.execute('pragma table_info(%s)' % view)]
    print(cols)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in cur.execute('select * from %s' % view):
        d = dict(zip(cols, row))
        k = re.sub(r'\(.*', '', d.get('kernel_name', d.get('name', '?')))[:70]
        acc[k][d.get('counter_name')].append(d.get('value', d.get('counter_value')))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print('   %-32s n=%d mean=%.4g' % (c, len(v), sum(v) / len(v)))

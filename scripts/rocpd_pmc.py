"""Print per-kernel PMC counter averages from a rocprofv3 rocpd sqlite db.  usage: rocpd_pmc.py <db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else "bsmm"
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name"
try:
    rows = db.execute(q, ("%" + sub + "%",)).fetchall()
except Exception as ex:
    print("schema:", cols, ex)
    sys.exit(1)
cur = None
for k, c, v, n in rows:
    if k != cur:
        print("==", k[:120], "(n=%d)" % n)
        cur = k
    print("   %-36s %18.1f" % (c, v))

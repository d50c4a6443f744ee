"""Last N kernel dispatches of a rocprofv3 kernel-trace database: start (us, relative), duration, queue, name."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
q = "queue_id" if "queue_id" in cols else cols[0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.%s from %s d join %s s on d.kernel_id=s.id order by d.start" % (q, kd, ks)).fetchall()
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[-(n + skip):len(rows) - skip]
t0 = rows[0][1]
for k, s, e, qq in rows:
    k = re.sub(r'^_ZN\d+_GLOBAL__N_1', '', k)
    print("%9.1f %8.1f q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qq, k[:80]))

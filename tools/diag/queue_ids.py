"""Distinct queue ids (and dispatch counts) of the k_gemm* / k_env_step dispatches in a rocprofv3 kernel-trace database, and how
many of the GEMM dispatches overlap another kernel in time."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
rows = rows[len(rows) // 2:]
c = collections.Counter((q, st) for k, s, e, q, st in rows)
print("(queue, stream) -> dispatches:", dict(c))
ov = sum(1 for i in range(1, len(rows)) if rows[i][1] < rows[i - 1][2])
print("dispatches starting before the previous one ended: %d of %d" % (ov, len(rows)))

import sqlite3, sys, collections, re
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# take last 40% of the rows (steady-state replays)
n=len(rows); rows=rows[int(n*0.6):]
span=(rows[-1][2]-rows[0][1])/1e3
agg=collections.defaultdict(lambda:[0,0.0])
for i,(k,s,e) in enumerate(rows):
    nxt = rows[i+1][1] if i+1<len(rows) else e
    k=re.sub(r'^_ZN\d+_GLOBAL__N_1','',k)[:90]
    agg[k][0]+=1; agg[k][1]+=(e-s)/1e3
tot=sum(v[1] for v in agg.values())
print("rows",len(rows),"span us",span,"busy us",tot)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:45]:
    print("%6d %9.1f %5.1f%% %6.2f  %s"%(v[0],v[1],100*v[1]/tot,v[1]/v[0],k))

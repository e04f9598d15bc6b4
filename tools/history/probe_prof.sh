cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/cp
rocprofv3 --kernel-trace -d gpurun_out/cp -o t -- python $1 > gpurun_out/cp/log.txt 2>&1
DB=$(find gpurun_out/cp -name "*.db" | head -1)
python - <<PY
import sqlite3,re
db=sqlite3.connect("$DB")
rows=db.execute("select name,start,end from kernels order by start").fetchall()
# group sequentially: names in call order; report per (name, occurrence-block) min
import collections
agg=collections.OrderedDict()
stage=0; seen=set(); last=None
for n,s,e in rows:
    n=re.sub(r'\(anonymous namespace\)::|void ','',n); n=re.sub(r'\(.*$','',n)[:60]
    if 'randn' in n or 'normal' in n or 'distribution' in n: 
        continue
    agg.setdefault(n,[]).append((e-s)/1e3)
for n,v in agg.items():
    # 4 stages x 10 reps each, in order
    k=len(v)//4 if len(v)>=4 else 1
    parts=[v[i*k:(i+1)*k] for i in range(4)] if len(v)>=4 else [v]
    print(f"{n:62s} n={len(v):3d} " + " ".join(f"{min(p):7.1f}/{sorted(p)[len(p)//2]:7.1f}" for p in parts if p))
PY
rm -f $DB

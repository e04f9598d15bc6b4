#!/usr/bin/env python3
"""Main stream vs weight-gradient side stream inside the last steps of a rocprofv3 kernel trace of bench.py: busy time of each (union of
its kernel intervals), time only ONE of them has a kernel in flight, and the inflation of the main-stream kernels while the side stream
runs.  Kernels are assigned by name (weight-gradient GEMMs, slab / partial-row reduces, dwconv weight gradient = side).
usage: two_stream.py results.db bench.log [nsteps]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
step_ms = float(re.search(r'ms_per_step": ([0-9.]+)', open(sys.argv[2]).read()).group(1))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = [c for c in cols if c.lower() in ("queue_id", "queue", "stream_id", "stream")]
tmax = db.execute("select max(end) from kernels").fetchone()[0]
rows = db.execute(f"select name, start, end{', ' + qcol[0] if qcol else ''} from kernels where start >= {int(tmax - n * step_ms * 1e6)} order by start").fetchall()
def is_side(nm):
    return bool(re.search(r"gemm_kernel<unsigned short, true, true, true|splitk_reduce|partial_reduce|reduce_batch|dwconv_bwd_w", nm))
def union(iv):
    iv = sorted(iv); out = []; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: out.append((cs, ce)); cs, ce = s, e
        else: ce = max(ce, e)
    out.append((cs, ce)); return out
def length(u): return sum(e - s for s, e in u)
def inter(a, b):
    i = j = 0; t = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s: t += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return t
side = union([(r[1], r[2]) for r in rows if is_side(r[0])])
main = union([(r[1], r[2]) for r in rows if not is_side(r[0])])
both = inter(side, main)
t0, t1 = rows[0][1], max(r[2] for r in rows)
ms = lambda x: x / n / 1e6
print(f"window {ms(t1 - t0):.2f} ms/step | main busy {ms(length(main)):.2f} | side busy {ms(length(side)):.2f} | both {ms(both):.2f} | only main {ms(length(main) - both):.2f} | only side {ms(length(side) - both):.2f} | neither {ms((t1 - t0) - length(main) - length(side) + both):.2f}")
if qcol: print("queues:", sorted(set(r[3] for r in rows)))
# the stretch of each step where side-stream kernels exist = the backward pass: how long is it, and how much of it has the main stream idle?
sb, se = side[0][0], side[-1][1]
# where in the step is the main stream idle while the side stream works?  (gaps of the main stream's busy union that overlap side kernels)
def sub(a, b):            # parts of intervals a not covered by b
    out = []; j = 0
    for s, e in a:
        cur = s
        while j < len(b) and b[j][1] <= cur: j += 1
        k = j
        while k < len(b) and b[k][0] < e:
            if b[k][0] > cur: out.append((cur, min(b[k][0], e)))
            cur = max(cur, b[k][1]); k += 1
        if cur < e: out.append((cur, e))
    return out
only_side = sub(side, main)
# step boundaries: the adamw kernel ends a step
ends = [r[2] for r in rows if "adamw_kernel" in r[0]]
print("longest main-idle / side-busy stretches (us, position in the step as ms before the optimizer launch):")
for s, e in sorted(only_side, key=lambda x: x[0] - x[1])[:12]:
    nxt = min([t for t in ends if t >= e], default=None)
    print(f"  {(e - s) / 1e3:8.1f} us   {((nxt - e) / 1e6 if nxt else float('nan')):7.2f} ms before the end of its step")

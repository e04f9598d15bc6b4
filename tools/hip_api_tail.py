#!/usr/bin/env python3
"""Aggregate a rocprofv3 hip_api_trace.csv over its last `ms` milliseconds (steady state): calls / total / max per API, plus the calls longer than 200 us with their time offset.
Usage: hip_api_tail.py trace_hip_api_trace.csv [ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
sk = "Start_Timestamp" if "Start_Timestamp" in rows[0] else [k for k in rows[0] if "tart" in k][0]
ek = "End_Timestamp" if "End_Timestamp" in rows[0] else [k for k in rows[0] if "nd_" in k or "End" in k][0]
nk = "Function" if "Function" in rows[0] else [k for k in rows[0] if "unction" in k or "Name" in k][0]
tmax = max(int(r[ek]) for r in rows if "LaunchKernel" in r[nk])          # the last launch (the tear-down after it is not the step)
t0 = tmax - ms * 1e6
agg = {}
long_calls = []
for r in rows:
    s, e = int(r[sk]), int(r[ek])
    if s < t0:
        continue
    a = agg.setdefault(r[nk], [0, 0, 0])
    a[0] += 1; a[1] += e - s; a[2] = max(a[2], e - s)
    if e - s > 200e3:
        long_calls.append((round((s - t0) / 1e6, 3), r[nk], round((e - s) / 1e3, 1), r.get("Thread_Id", "")))
print("api,calls,total_ms,max_us")
for k, (c, t, m) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k},{c},{t / 1e6:.3f},{m / 1e3:.1f}")
print("calls > 200 us (offset ms, api, us, thread):")
for x in long_calls[:60]:
    print(x)

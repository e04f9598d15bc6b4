"""Per-phase instruction statistics of a kernel's ISA: phases are the `; PHASE_<name>` asm markers of the source.
usage: python tools/isa_phases.py file.s kernel_substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if key in l and l.split(';')[0].strip().endswith(':')][0]
end = [i for i, l in enumerate(lines) if i > start and '.Lfunc_end' in l and l.strip().endswith(':')][0]
kinds = ['scratch_load', 'scratch_store', 'v_mfma', 'ds_read', 'ds_write', 'ds_bpermute', 'global_load', 'buffer_load', 'buffer_store', 's_waitcnt', 's_barrier', 'v_readlane', 'v_writelane', 's_nop', 'v_exp', 'v_pk_']
phase, cnt, order = 'PRE', {}, []
for l in lines[start:end]:
    m = re.search(r'; PHASE_(\w+)', l)
    if m: phase = m.group(1)
    if phase not in cnt: cnt[phase] = dict(insts=0, **{k: 0 for k in kinds}); order.append(phase)
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.split(';')[0].strip().endswith(':'): continue
    cnt[phase]['insts'] += 1
    for k in kinds:
        if t.startswith(k): cnt[phase][k] += 1
for p in order:
    print(f"{p:10s}", ' '.join(f"{k}={v}" for k, v in cnt[p].items() if v))

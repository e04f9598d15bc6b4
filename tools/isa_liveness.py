"""Crude straight-line liveness profile of a kernel's ISA (ignores branches): max simultaneously live VGPR/AGPRs and a profile every N instructions.
usage: python tools/isa_liveness.py file.s kernel_substring [step]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]; step = int(sys.argv[3]) if len(sys.argv) > 3 else 150
st = [i for i, l in enumerate(lines) if key in l and l.split(';')[0].strip().endswith(':')][0]
def regs(tok):
    out = []
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', tok):
        if m.group(1): out += [(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)]
        else: out.append((m.group(4), int(m.group(5))))
    return out
ins = []
for l in lines[st + 1:]:
    if '.Lfunc_end' in l: break
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'): continue
    parts = t.split(None, 1)
    op = parts[0]; args = parts[1] if len(parts) > 1 else ''
    ops = [a.strip() for a in args.split(',')]
    if op.startswith(('ds_write', 'buffer_store', 'global_store', 'scratch_store', 's_')):
        d = []; s = [r for o in ops for r in regs(o)]
    else:
        d = regs(ops[0]) if ops else []; s = [r for o in ops[1:] for r in regs(o)]
    ins.append((t, d, s))
live = set(); prof = []
for t, d, s in reversed(ins):
    for r in d: live.discard(r)
    for r in s: live.add(r)
    prof.append(len(live))
prof = prof[::-1]
print("instructions", len(ins), "max live", max(prof))
for i in range(0, len(ins), step): print(i, prof[i], ins[i][0][:70])

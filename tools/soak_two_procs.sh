#!/bin/bash
# Two processes on ONE device, each launching the persistent stage kernels as if it owned the chip (the residency bound counts the launches of one process: DESIGN 4.10).
# Acceptable outcomes: both finish with bit-identical repeats and error word 0, or a process raises (a lost hand-off is an exception) -- never silently different outputs.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/two_procs; mkdir -p $O
cat > /tmp/soak_one.py <<'P'
import sys, torch
sys.path.insert(0, ".")
import lemevit_amd
from lemevit_amd import ops
from lemevit_amd.graph import split_forward
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = lemevit_amd.create_model("lemevit_base", num_classes=1000).to(dev).eval()
x = torch.randn(128, 3, 224, 224, device=dev)
outs, first, diff, raised = [], None, 0, 0
with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
    for i in range(150):
        outs.clear()
        try:
            split_forward(model, x, 1, outs)
            o = outs[0].float()
            torch.cuda.synchronize()
            ops.check_stage_errors("two procs", sync=True)
        except RuntimeError as e:
            raised += 1
            if raised == 1: print("raised:", str(e)[:160], flush=True)
            ops.stage_kernels_disabled = False if hasattr(ops, "stage_kernels_disabled") else None
            continue
        if first is None: first = o.clone()
        elif not torch.equal(o, first): diff += 1
print(f"process {sys.argv[1]}: 150 passes, {raised} raised, {diff} silent differences, finite {first is not None and bool(torch.isfinite(first).all())}", flush=True)
P
timeout 280 python /tmp/soak_one.py A > $O/a.log 2>&1 &
PA=$!
timeout 280 python /tmp/soak_one.py B > $O/b.log 2>&1 &
PB=$!
wait $PA; echo "A rc=$?"; wait $PB; echo "B rc=$?"
grep -v amdgpu $O/a.log | tail -3; grep -v amdgpu $O/b.log | tail -3

import sys; sys.path.insert(0, "/root/repo")
import torch, collections
import lemevit_amd
m = lemevit_amd.create_model("lemevit_base", num_classes=1000).to("cuda:0").eval()
x = torch.randn(128, 3, 224, 224, device="cuda:0")
with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m(x); torch.cuda.synchronize()
cnt = collections.Counter()
evs = list(prof.events())
print(len(evs))
names = collections.Counter(e.name for e in evs)
print([ (k, v) for k, v in names.items() if "opy" in k or "emcpy" in k or "emset" in k])
for ev in evs:
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::zero_", "aten::fill_", "aten::cat") :
        fr = [s for s in (ev.stack or []) if "lemevit_amd" in s][:2]
        cnt[(ev.name, tuple(fr))] += 1
for k, v in cnt.most_common(30):
    print(v, k)

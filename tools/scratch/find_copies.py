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
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy") and ev.stack:
        fr = [s for s in ev.stack if "lemevit_amd" in s][:2]
        cnt[(ev.name, tuple(fr))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
print([ (e.key, e.count) for e in prof.key_averages() if "emcpy" in e.key or "copyBuffer" in e.key])

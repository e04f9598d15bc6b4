#!/usr/bin/env python3
"""cProfile of the host side of the Base train step (10 steps, eager): where the Python / ctypes time per step goes."""
import cProfile, pstats, sys, torch
sys.path.insert(0, ".")
import lemevit_amd
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = lemevit_amd.create_model("lemevit_base", num_classes=1000, drop_path_rate=0.1).to(dev).train()
opt = lemevit_amd.FlatAdamW(model, lr=1e-4, eps=1e-8, weight_decay=0.05)
x = torch.randn(128, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (128,), device=dev)
lossf = torch.nn.CrossEntropyLoss()
def step():
    with torch.autocast("cuda", torch.bfloat16):
        loss = lossf(model(x), y)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=False)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 40)

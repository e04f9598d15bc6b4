#!/usr/bin/env python3
"""cProfile of the host side of eager train steps (where does the 32 ms of launch work per step go?)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lemevit_amd
dev = torch.device("cuda:0")
model = lemevit_amd.create_model("lemevit_base", num_classes=1000, drop_path_rate=0.1).to(dev).train()
opt = lemevit_amd.FlatAdamW(model, lr=1e-4, weight_decay=0.05)
x = torch.randn(128, 3, 224, 224, device=dev); lf = torch.nn.CrossEntropyLoss()
def step():
    opt.zero_grad()
    with torch.autocast("cuda", torch.bfloat16):
        lf(model(x), torch.empty((128,), device=dev, dtype=torch.long).random_(1000)).backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime")
st.print_stats(22)

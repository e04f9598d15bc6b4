import os, torch, torch.distributed as dist, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import lemevit_amd
from lemevit_amd.dist import wrap_ddp
torch.manual_seed(0)
m = lemevit_amd.create_model("lemevit_tiny", num_classes=10).cuda().train()
x = torch.randn(8, 3, 96, 96, device="cuda"); y = torch.randint(0, 10, (8,), device="cuda")
def grads(net, base):
    base.zero_grad(set_to_none=True)
    with torch.autocast("cuda", torch.bfloat16):
        torch.nn.functional.cross_entropy(net(x), y).backward()
    return {n: p.grad.clone() for n, p in base.named_parameters()}
g0 = grads(m, m); g1 = grads(m, m)
def cmp(a, b, tag):
    worst = sorted(((float((a[n] - b[n]).abs().max() / (b[n].abs().max() + 1e-12)), n) for n in a), reverse=True)[:4]
    print(tag, [(round(v, 5), n) for v, n in worst])
cmp(g0, g1, "plain run-to-run:")
for bf in (False, True):
    m2 = copy.deepcopy(m)
    d = wrap_ddp(m2, 0, bf16_grads=bf)
    cmp(grads(d, m2), g1, f"DDP bf16_grads={bf} vs plain:")
dist.destroy_process_group()

"""What would the saved set of a training form cost inside the persistent S-stage launch?    Needs the -DSS_DBG_SAVE=1 build of csrc/sstage.hip:
    LMV_LIB_PATH=.../liblemevit_hip_SAVE.so python tools/ss_save_cost.py [B=128] [nblocks=18]
The kernel then also writes, per block, what lmv_block_bwd reads back -- LayerNorm inputs and outputs, packed qkv, attention output, t2, u, h: 16 C per token, row-major, in the
8-byte (v: 2-byte) pieces its fragment layouts give -- into a buffer of its own per block (nothing is overwritten: the stores go to HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from lemevit_amd import ops
from test_sstage_gpu import _inputs, _pack, _stage_params, DEV, G
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 18
C = 384
P = _pack(_stage_params(nb, 9, C))
x, c = _inputs(B, 4, C=C)
x, c = x.to(DEV), c.to(DEV)
nwg = 2 * ((B + 7) // 8) * 8
per = 16 * 112 * C * 2
save = torch.empty(nb * nwg * per, device=DEV, dtype=torch.uint8)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
plain = t(lambda: ops.sstage_fwd(x, c, P, G, G, 1e-6))
saving = t(lambda: ops.sstage_fwd(x, c, P, G, G, 1e-6, timing=save, timing_block=-7))
a = ops.sstage_fwd(x, c, P, G, G, 1e-6); b = ops.sstage_fwd(x, c, P, G, G, 1e-6, timing=save, timing_block=-7)
torch.cuda.synchronize()
gb = nb * nwg * per / 1e9
print(f"lib {os.environ.get('LMV_LIB_PATH', 'default')}: B = {B}, {nb} blocks: {plain:.3f} ms without, {saving:.3f} ms with the saved set ({gb:.2f} GB = {gb / (saving * 1e-3) / 1e3:.2f} TB/s of stores); "
      f"outputs equal: {torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])}; a written word: {int(save[12345])}")

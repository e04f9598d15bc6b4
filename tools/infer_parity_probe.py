import sys, os, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden"); sys.path.insert(0, "tests")
import lemevit_amd
from detfill import det_tensor, fill_state_dict
from oracle import lemevit_oracle as O
import lemevit_amd.model as M
dev = "cuda:0"
for variant, res in (("lemevit_tiny", 96), ("lemevit_tiny", 224), ("lemevit_base", 224)):
    cfg = O.VARIANTS[variant]
    torch.manual_seed(0)
    model = lemevit_amd.create_model(variant, num_classes=10).to(dev).eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    img = det_tensor((2, 3, res, res), "s.img", 3)
    ref = O.lemevit_forward(sd, cfg, img)
    out = {}
    for name, fused, stage, dst in (("stage kernels", True, True, True), ("no sstage", True, False, True), ("no dstage", True, True, False), ("fused per-launch", True, False, False),
                                    ("unfused per-launch", False, False, False)):
        M._FUSED, M._SSTAGE, M._DSTAGE = fused, stage, dst
        with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
            o = model(img.to(dev)).float().cpu()
        out[name] = float((o - ref).abs().max() / ref.abs().max())
    M._FUSED, M._SSTAGE, M._DSTAGE = True, True, True
    print(variant, res, "reference-init weights:", {k: f"{v:.2e}" for k, v in out.items()})

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from litehandnet_amd import get_loss, get_model
from litehandnet_amd.config import litehandnet_cfg
from oracle import heatmap_np as onp, synth, torch_ref
variant, size, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), 3
cfg = litehandnet_cfg(variant); cfg.MODEL["ca_dropout"] = 0.0
ref = torch_ref.get_model(cfg, p_drop=0.0); sd = synth.synth_state_dict(ref, seed); ref.load_state_dict(sd); ref = ref.double().train()
hs = size // 4
j = synth.synth_joints(n, 21, size, seed + 1)
tgt = np.stack([onp.msra_generate_target(a, np.ones_like(a), [size, size], [hs, hs])[0] for a in j]); tw = torch.ones(n, 21, 1)
x = synth.synth_images(n, size, seed)
y64 = ref(x.double()); l64 = torch_ref.distance_loss(y64, torch.from_numpy(tgt).double(), tw.double()); l64.backward()
m = get_model(cfg); m.load_state_dict(sd); m.cuda().train()
y = m(x.cuda()); loss, _ = get_loss(cfg)(y, {"target": torch.from_numpy(tgt), "target_weight": tw}); loss.backward()
print("fwd err", float((y.detach().cpu().double() - y64.detach()).abs().max() / y64.abs().max()))
rp = dict(ref.named_parameters()); floor = 1e-3 * max(float(v.grad.norm()) for v in rp.values())
errs = sorted(((float((p.grad.cpu().double() - rp[k].grad).norm() / (rp[k].grad.norm() + floor)), k) for k, p in m.named_parameters()), reverse=True)
for e, k in errs[:12]: print(f"{e:.3e} {k}")

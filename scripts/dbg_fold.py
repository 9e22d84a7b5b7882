import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import torch_ref, synth
from litehandnet_amd.repblocks import RepConv
ours, ref = RepConv(32, 64, 1), torch_ref.RepConv(32, 64, 1)
sd = synth.synth_state_dict(ref, 40)
ref.load_state_dict(sd); ours.load_state_dict(sd)
ours.cuda().eval(); ref.eval()
ours.switch_to_deploy(); ref.switch_to_deploy()
a = ours.rep_conv.weight.cpu().numpy().reshape(64, 32); b = ref.rep_conv.weight.numpy().reshape(64, 32)
print("w mismatches", (a != b).sum(), "of", a.size, "max ulp-ish", np.abs(a - b).max() / np.abs(b).max())
print("b mismatches", (ours.rep_conv.bias.cpu().numpy() != ref.rep_conv.bias.numpy()).sum())
g, rv, w = sd["conv.bn.weight"].numpy(), sd["conv.bn.running_var"].numpy(), sd["conv.conv.weight"].numpy().reshape(64, 32)
std = np.sqrt(rv + np.float32(1e-5)); t = g / std
n = w * t[:, None]
print("numpy vs torch", (n != b).sum(), "numpy vs gpu", (n != a).sum())
rows = np.where((a != b).any(1))[0]
print("rows", rows[:10], "t gpu-implied", (a[rows[0], 0] / w[rows[0], 0]), t[rows[0]])

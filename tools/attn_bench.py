"""Attention core (fp16x3 flash kernel) alone at the denoisers' size: time per call + a checksum + distance from a float64 softmax.
usage: attn_bench.py [B=16] [C=64] [L=4096]   (A/B: BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdm_amd import ops

B, C, Lk = (int(v) for v in (sys.argv[1:4] + ["16", "64", "4096"][len(sys.argv) - 1:]))
g = torch.Generator().manual_seed(3)
qkv = (torch.randn(B, 3 * C, Lk, generator=g) * 0.7).cuda()
amax = torch.stack([qkv[:, i * C:(i + 1) * C].abs().amax(dim=(1, 2)) for i in range(3)], dim=1).contiguous()   # (B, 3): max |q|, |k|, |v| per shape
out = ops.attention_core(qkv, C, amax=amax)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    s.record()
    for _ in range(10):
        out = ops.attention_core(qkv, C, amax=amax)
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 100)
q, k, v = (qkv[:2, i * C:(i + 1) * C].double() for i in range(3))
ref = torch.einsum("bcj,bij->bci", v, torch.softmax(torch.einsum("bci,bcj->bij", q, k), dim=2))
err = float((out[:2].double() - ref).norm() / ref.norm())
print(f"attention_core B={B} C={C} L={Lk}: {min(ts):.1f} us per call (3 launches; min of 5 x 10), checksum {float(out.double().sum()):.9e}, rel-L2 vs float64 {err:.2e}")

#!/usr/bin/env python
"""Does a long chain of ANOTHER shape remove the first-call cost of a new length (runtime pools that grow with the number of
graph launches in flight)?  usage: python tools/first_call2.py [A|B]"""
import sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(2021)
mk = lambda n: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), n).items()}
b8, b2 = mk(8), mk(2)
def call(b, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.sample_diffusion(num_steps=n, center_pos_mode="protein", seed=n, **b)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
mode = sys.argv[1] if len(sys.argv) > 1 else "A"
print(mode, "warm-up 5 (B=8):", round(call(b8, 5), 1))
if mode == "B":
    print("  other shape (B=2) 5 steps:", round(call(b2, 5), 1), " 100 steps:", round(call(b2, 100), 1))
for i in range(3 if mode != "C" else 0):
    print("  20 steps (B=8):", round(call(b8, 20), 2))
if mode == "C":     # sustained heavy load right before: does the next 20-step call run at the steady rate?
    a = torch.randn(8192, 8192, device=dev)
    for ms in (30, 100):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        while time.perf_counter() - t0 < ms * 1e-3:
            (a @ a).sum().item()
        print(f"  after {ms} ms of matmul: 20 steps (B=8):", round(call(b8, 20), 2), " then:", round(call(b8, 20), 2))
        time.sleep(0.5)
        print("  after 0.5 s idle: 20 steps:", round(call(b8, 20), 2), " then:", round(call(b8, 20), 2))

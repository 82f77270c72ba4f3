#!/usr/bin/env python
"""Where does the first 20-step call after a 5-step warm-up spend its extra ~1.7 ms?  Phase timers (host clock, device
synchronised at each boundary) around _prepare_chain / _run_chain_streaming / _collect_chain and the streaming internals."""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth, hip_lib
import decompdiff_amd.model as M
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(2021)
b8 = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), 8).items()}
T = {}
def wrap(name):
    f = getattr(DecompScorePosNet3D, name)
    def g(self, *a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(self, *a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0) + 1e3 * (time.perf_counter() - t0)
        return r
    setattr(DecompScorePosNet3D, name, g)
for n in ("_prepare_chain", "_run_chain_streaming", "_collect_chain"):
    wrap(n)
lib = hip_lib.load()
orig_launch = lib.dd_graph_launch
EV = []
class L:
    def __call__(self, *a):
        st = torch.cuda.ExternalStream(a[2])
        if not EV:
            e = torch.cuda.Event(enable_timing=True); e.record(st); EV.append(e)
        t0 = time.perf_counter(); r = orig_launch(*a); T["launch_host"] = T.get("launch_host", 0) + 1e3 * (time.perf_counter() - t0)
        e = torch.cuda.Event(enable_timing=True); e.record(st); EV.append(e)
        return r
lib.dd_graph_launch = L()
def call(n):
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.sample_diffusion(num_steps=n, center_pos_mode="protein", seed=n, **b8)
    torch.cuda.synchronize(); tot = 1e3 * (time.perf_counter() - t0)
    T["gpu_chain"] = EV[0].elapsed_time(EV[-1]); T["pieces"] = "/".join(f"{EV[i].elapsed_time(EV[i+1]):.2f}" for i in range(len(EV) - 1)); EV.clear()
    return f"{n:3d} steps: total {tot:6.2f}  " + "  ".join(f"{k} {v:.2f}" if not isinstance(v, str) else f"{k} {v}" for k, v in T.items())
for a in (sys.argv[1:] or "5 20 20 20 20 40 40 20".split()):
    if a == "M":      # grow glibc's mmap threshold and leave a touched, freed heap region behind
        x = torch.empty(32 << 20, dtype=torch.uint8).fill_(1); del x
        y = [torch.empty(2 << 20, dtype=torch.uint8).fill_(1) for _ in range(8)]; del y
        print("malloc warmed")
    elif a == "S":
        from decompdiff_amd.dist import device_spin
        device_spin(dev, 150.0); print("spun 150 ms")
    else:
        print(call(int(a)))

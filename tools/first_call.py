#!/usr/bin/env python
"""Is the first sample_diffusion call of a given length slower than the following ones, and why?  usage: python tools/first_call.py"""
import sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(2021)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), 8).items()}
def call(n, keep=True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.sample_diffusion(num_steps=n, center_pos_mode="protein", seed=n, keep_traj=keep, **b)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0), out
t, _ = call(5); print(f"warm-up 5 steps: {t:.1f} ms")
held = []
for n in (20, 20, 20, 40, 40, 20, 10, 10):
    t, out = call(n); held.append(out)                     # results kept alive: every call allocates fresh host memory
    print(f"{n:3d} steps (results kept): {t:.2f} ms = {t / n:.3f} ms/step")
held.clear()
for n in (20, 20, 20):
    t, out = call(n)
    print(f"{n:3d} steps (results dropped): {t:.2f} ms")
for n in (20, 20):
    t, out = call(n, keep=False)
    print(f"{n:3d} steps (no trajectories): {t:.2f} ms")
# GPU time of the chain itself (events on the chain's stream) vs the host-visible time, first and later calls of a new length
import types
orig = m._run_chain_streaming
def timed(self, chain, num_steps):
    side = self._side_stream(chain["dev"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side.wait_stream(torch.cuda.current_stream(chain["dev"]))
    e0.record(side)
    t0 = time.perf_counter()
    r = orig(chain, num_steps)
    t1 = time.perf_counter()
    e1.record(side); e1.synchronize()
    print(f"    chain of {num_steps}: GPU {e0.elapsed_time(e1):.2f} ms, host in _run_chain_streaming {1e3 * (t1 - t0):.2f} ms")
    return r
m._run_chain_streaming = types.MethodType(timed, m)
for n in (24, 24, 24, 28, 28):
    t, out = call(n)
    print(f"{n:3d} steps: {t:.2f} ms")

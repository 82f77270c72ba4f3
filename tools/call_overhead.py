#!/usr/bin/env python
"""Fixed cost of one short sample_diffusion call (the driver's bench times 20 steps): wall time of each host phase,
with device syncs between them.  usage: python tools/call_overhead.py [steps] [reps]"""
import sys, time, ctypes, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), 8).items()}
m.sample_diffusion(num_steps=5, center_pos_mode="protein", seed=1, **b); torch.cuda.synchronize()
sync = torch.cuda.synchronize
T = {}
def timed(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        sync(); t0 = time.perf_counter(); r = f(*a, **k); sync(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, w)
for n in ("_is_ragged", "_dense_inputs", "_make_sampler", "_prepare_chain", "_run_chains", "_collect_chain", "_packed_weights"):
    timed(m, n)
lib = hip_lib.load()
for n in ("dd_graph_create", "dd_graph_launch", "dd_graph_destroy", "dd_embed_protein"):
    f = getattr(lib, n)
    def mk(f, n):
        def w(*a):
            t0 = time.perf_counter(); r = f(*a); T[n] = T.get(n, 0.0) + time.perf_counter() - t0; return r
        return w
    setattr(lib, n, mk(f, n))
for rep in range(reps):
    T.clear()
    sync(); t0 = time.perf_counter()
    m.sample_diffusion(num_steps=steps, center_pos_mode="protein", seed=2 + rep, **b)
    sync(); tot = time.perf_counter() - t0
    print(f"call {rep}: total {1e3*tot:.2f} ms for {steps} steps ({1e3*tot/steps:.3f} ms/step) | " +
          " | ".join(f"{k} {1e3*v:.2f}" for k, v in T.items()))
# without the instrumentation syncs
for rep in range(3):
    sync(); t0 = time.perf_counter()
    m.sample_diffusion.__wrapped__(m, num_steps=steps, center_pos_mode="protein", seed=9 + rep, **b) if hasattr(m.sample_diffusion, "__wrapped__") else m.sample_diffusion(num_steps=steps, center_pos_mode="protein", seed=9 + rep, **b)
    sync(); print(f"plain call: {1e3*(time.perf_counter()-t0):.2f} ms")

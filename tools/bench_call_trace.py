#!/usr/bin/env python
"""The timed call of `bench.py --steps 20 --warmup 5` with wall-clock stamps around the phases of sample_diffusion (no extra
syncs): where do the ~2 ms on top of 20 x 1.2 ms go?  usage: python tools/bench_call_trace.py [gc 0|1]"""
import gc, sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
from decompdiff_amd import dist as ddist
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, seed=0)); m.load_state_dict(sd, strict=True); m = m.to(dev)
u = ddist.plan_job(1, 1)[0][0]
pocket = synth.make_pocket(u.pocket_seed, u.num_protein, u.arm_atoms, u.scaffold_atoms, num_full_protein=0)
torch.manual_seed(u.init_seed)
bc = synth.build_sampling_batch(pocket, u.n_samples, per_sample_std_scale=[1.0] * u.n_samples)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in bc.items()}
T = []
def stamp(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.append((name, 1e3 * (time.perf_counter() - t0))); return r
    setattr(obj, name, w)
for n in ("_static_memo_get", "_prepare_chain", "_make_sampler", "_run_chains", "_run_chain_streaming", "_collect_chain", "_packed_weights"):
    if hasattr(m, n): stamp(m, n)
def sample(n, seed):
    return m.sample_diffusion(num_steps=n, center_pos_mode="protein", energy_drift_opt=None, seed=seed, keep_traj=True, use_graph=True, **b)
ddist.checksum(sample(5, u.noise_seed + 1))
if len(sys.argv) < 2 or sys.argv[1] == "1":
    gc.collect(); gc.disable()
for rep in range(4):
    T.clear()
    m.__dict__["_trace"] = tr = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = sample(20, u.noise_seed + rep)
    t1 = time.perf_counter(); ddist.checksum(out); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"call {rep}: sample {1e3*(t1-t0):.2f} ms + checksum {1e3*(t2-t1):.2f} | " + " | ".join(f"{k} {v:.2f}" for k, v in T))
    print("    streaming stamps (ms since call start): " + "  ".join(f"{k} {1e3*(t-t0):.2f}" for k, t in tr))
m.__dict__["_trace"] = None
# the device's own view: events around the 20 graph replays of a steady call
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side = m._side_stream(dev)
    torch.cuda.synchronize(); e0.record(side)
    out = sample(20, u.noise_seed + 10 + rep)
    e1.record(side); torch.cuda.synchronize()
    print(f"side-stream events around the call: {e0.elapsed_time(e1):.2f} ms")

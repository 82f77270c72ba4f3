import sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
"""Step time for ligand sizes off the bench shape: mid-size (3-tile kernels), 60 (4-tile) and beyond 64 atoms (8-tile kernels,
register spills accepted).  usage: python tools/mid_size_time.py"""
for nl_arms, sca, np_ in (((15, 15), 15, 450), ((12, 12), 12, 400), ((20, 20), 20, 500), ((27, 27), 26, 500), ((32, 32), 32, 500), ((43, 43), 42, 500)):
    pocket = synth.make_pocket(0, np_, nl_arms, sca); torch.manual_seed(0)
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, 8).items()}
    def run(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.sample_diffusion(num_steps=steps, center_pos_mode="protein", keep_traj=True, use_graph=True, **b)
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps
    run(20)
    n = 150 if sum(nl_arms) + sca <= 64 else 40
    print(f"NL={sum(nl_arms)+sca} NP={np_} B=8: {min(run(n) for _ in range(3)):.4f} ms/step", flush=True)

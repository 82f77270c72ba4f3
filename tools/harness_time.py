import sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth, harness
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
pocket = synth.make_pocket_small(0)
harness.sample_diffusion_ligand_decomp(m, pocket, 8, 8, device="cuda:0", num_steps=20)
for rep in range(2):
    t0 = time.perf_counter()
    out = harness.sample_diffusion_ligand_decomp(m, pocket, 16, 8, device="cuda:0", num_steps=1000)
    t1 = time.perf_counter()
    recs = harness.to_result_records(out)
    t2 = time.perf_counter()
    print(f"harness 2 batches x B=8 x 1000 steps: {t1-t0:.3f} s total, model time {sum(out['time_list']):.3f} s, records {t2-t1:.3f} s")

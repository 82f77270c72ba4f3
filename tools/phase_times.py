#!/usr/bin/env python
"""Host-side phases of one sample_diffusion call (prepare / graph capture+replay / collect), shipped workload.
usage: python tools/phase_times.py [steps]"""
import sys, time, ctypes, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), 8).items()}
m.sample_diffusion(num_steps=20, center_pos_mode="protein", **b); torch.cuda.synchronize()
sync = torch.cuda.synchronize
for rep in range(2):
    t0 = time.perf_counter()
    chain = m._prepare_chain(b["protein_pos"], b["protein_v"], b["batch_protein"], b["init_ligand_pos"], b["init_ligand_v"],
                             b["ligand_v_aux"], b["batch_ligand"], b["prior_stds"], b["ligand_decomp_batch"], b["ligand_decomp_index"],
                             None, b["ligand_fc_bond_index"], b["init_ligand_fc_bond_type"], steps, "protein", None, None, None,
                             None, 0, True, 0)
    sync(); t1 = time.perf_counter()
    m._run_chains([chain], steps, True)
    sync(); t2 = time.perf_counter()
    out = m._collect_chain(chain, steps, True)
    sync(); t3 = time.perf_counter()
    print(f"prepare {1e3*(t1-t0):7.2f} ms | run {1e3*(t2-t1):8.2f} ms ({1e3*(t2-t1)/steps:.4f} ms/step) | collect {1e3*(t3-t2):7.2f} ms | total/step {1e3*(t3-t0)/steps:.4f}")

# host cost of replaying the step graph: time for the launch calls to return vs time until the GPU is done
lib = hip_lib.load()
for bb in (1, 8):
    torch.manual_seed(0)
    bx = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), bb).items()}
    chain = m._prepare_chain(bx["protein_pos"], bx["protein_v"], bx["batch_protein"], bx["init_ligand_pos"], bx["init_ligand_v"],
                             bx["ligand_v_aux"], bx["batch_ligand"], bx["prior_stds"], bx["ligand_decomp_batch"], bx["ligand_decomp_index"],
                             None, bx["ligand_fc_bond_index"], bx["init_ligand_fc_bond_type"], 1000, "protein", None, None, None,
                             None, 0, False, 0)
    st = torch.cuda.Stream(device=dev)
    g = ctypes.c_void_p()
    t0 = time.perf_counter()
    hip_lib.check(lib.dd_graph_create(ctypes.byref(chain["s"]), 1, st.cuda_stream, ctypes.byref(g)), "create")
    t1 = time.perf_counter()
    for n in (20, 200):
        sync(); ta = time.perf_counter()
        hip_lib.check(lib.dd_graph_launch(g, n, st.cuda_stream), "launch")
        tb = time.perf_counter(); sync(); tc = time.perf_counter()
        print(f"B={bb}: graph create {1e3*(t1-t0):.2f} ms; {n} replays: host returns after {1e3*(tb-ta)/n:.4f} ms/graph, GPU done after {1e3*(tc-ta)/n:.4f} ms/graph")
    lib.dd_graph_destroy(g)

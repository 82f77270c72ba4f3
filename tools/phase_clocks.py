#!/usr/bin/env python
"""Print s_memtime phase durations of the tiled attention kernels (profiling aid, needs a GPU)."""
import ctypes, sys, torch, numpy as np
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0")
cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
pocket = synth.make_pocket_small(0); torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, 8).items()}
lib = hip_lib.load()
names = ["stage weights/tables", "query + Q~ fold", "angle codes", "stage W2v (NE/PE)", "pass1 tiles", "softmax", "-", "pass2 tiles", "-", "epilogue"]
for mode, nm in enumerate(["NE", "NB", "BL", "PE", "PB"]):
    buf = torch.zeros(4096, 16, dtype=torch.int64, device=dev)
    lib.dd_debug_set_clock_buffer(ctypes.c_void_p(buf.data_ptr()), mode)
    m.sample_diffusion(num_steps=1, center_pos_mode="protein", keep_traj=False, use_graph=False, **b)
    torch.cuda.synchronize()
    lib.dd_debug_set_clock_buffer(None, -1)
    c = buf.cpu().numpy()
    c = c[c[:, 0] != 0]
    if nm == "BL" and len(c) and c[:, 12].all():            # cooperative persistent workgroups (bl_coop_body): 13 stamps per TRIP
        cn = ["coop fold", "barrier C", "Q~ read + barrier D", "angle codes", "k pass", "softmax", "v pass", "next trip's loads issued",
              "Z~ written", "barrier E", "epilogue MFMAs", "stores"]
        d = np.diff(c[:, :13], axis=1).astype(np.float64)
        tot = (c[:, 12] - c[:, 0]).astype(np.float64)
        print(f"BL (cooperative trips): {len(c)} trips x 6 layers (last layer's stamps), median trip {np.median(tot):.0f} ticks -> {np.median(tot)/2400:.1f} us")
        print("   " + " | ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(cn)))
        f = np.stack([c[:, 4], c[:, 13], c[:, 14], c[:, 15], c[:, 5]], 1).astype(np.float64)
        print("   k pass: segment row arrives %.0f | tile 0 table MFMAs + LayerNorm %.0f | tile 0 scores %.0f | tile 1 %.0f" % tuple(np.median(np.diff(f, axis=1), axis=0)))
        continue
    last = 10
    d = np.diff(c[:, :last + 1], axis=1).astype(np.float64)
    tot = (c[:, last] - c[:, 0]).astype(np.float64)
    print(f"{nm}: {len(c)} workgroups x 6 layers, median total {np.median(tot):.0f} ticks of s_memtime (shader clock, ~2.4 GHz -> {np.median(tot)/2400:.1f} us)")
    print("   " + " | ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names[:d.shape[1]])))
    f = np.concatenate([c[:, 4:5], c[:, 11:15]], axis=1).astype(np.float64)
    print("   tile0/pass1: rows arrive %.0f | table MFMAs %.0f | LayerNorm %.0f | score MFMAs %.0f" % tuple(np.median(np.diff(f, axis=1), axis=0)))

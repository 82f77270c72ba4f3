import sys, subprocess
CHILD = r'''
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import golden_utils as GU
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
g = GU.load("forward_small"); b = GU.batch_from_npz(g)
lib = hip_lib.load()
k, v = int(sys.argv[1]), int(sys.argv[2])
assert lib.dd_debug_set_option(k, v) == 0
kw = {kk: (vv.to(dev) if torch.is_tensor(vv) else vv) for kk, vv in b.items()}
names = ["protein_pos","protein_v","batch_protein","protein_group_idx","init_ligand_pos","init_ligand_v","batch_ligand","ligand_group_idx","prior_centers","prior_stds","batch_prior","prior_group_idx","ligand_fc_bond_index","init_ligand_fc_bond_type"]
o = m(init_ligand_v_aux=kw["ligand_v_aux"], **{n: kw.get(n) for n in names})
torch.cuda.synchronize(); print("ok", k, v, float(o["pred_ligand_pos"].abs().sum()))
'''
for k, v in ((1, 0), (3, 0), (5, 8), (8, 1), (8, 2), (9, 0), (11, 1), (12, 0)):
    r = subprocess.run([sys.executable, "-c", CHILD, str(k), str(v)], capture_output=True, text=True)
    print(k, v, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], (r.stderr.strip().splitlines() or [""])[-1][:120] if r.returncode else "")

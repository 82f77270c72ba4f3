"""GPU (-m gpu): the training / validation objective (models/decompdiff.py:419-550) against the
reference's own get_diffusion_loss + backward (tests/golden/loss_grad.npz, oracle/make_golden.py --only loss): losses,
network outputs, the gradient of a spread of parameters and the gradient norm of every parameter.  Tolerances: the
autograd path uses torch (rocBLAS) GEMMs and the HIP scatter ops in fp32 -- 1e-4 on outputs, 1e-3 relative on gradients."""
import numpy as np
import pytest
import torch

import golden_utils as GU
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
from test_gpu_parity import dev, maxabs

pytestmark = pytest.mark.gpu


def _fresh_model():
    cfg = shipped_config()
    m = DecompScorePosNet3D(cfg, 29, 10, 8)
    sd = m.state_dict()
    sd.update(synth.synthetic_state_dict(cfg, 0))
    m.load_state_dict(sd, strict=True)
    return m.to(dev())


def _loss_kwargs(g):
    b = GU.batch_from_npz(g)
    d = lambda t: t.to(dev()) if torch.is_tensor(t) else t
    return dict(protein_pos=d(b["protein_pos"]), protein_v=d(b["protein_v"]), batch_protein=d(b["batch_protein"]),
                protein_group_idx=d(b["protein_group_idx"]), ligand_pos=d(b["init_ligand_pos"]), ligand_v=d(b["init_ligand_v"]),
                ligand_v_aux=d(b["ligand_v_aux"]), batch_ligand=d(b["batch_ligand"]), ligand_group_idx=d(b["ligand_group_idx"]),
                prior_centers=d(b["prior_centers"]), prior_stds=d(b["prior_stds"]), prior_num_atoms=d(b["prior_num_atoms"]),
                batch_prior=d(b["batch_prior"]), prior_group_idx=d(b["prior_group_idx"]),
                ligand_decomp_batch=d(b["ligand_decomp_batch"]), ligand_decomp_index=d(b["ligand_decomp_index"]),
                ligand_fc_bond_index=d(b["ligand_fc_bond_index"]), ligand_fc_bond_type=d(b["init_ligand_fc_bond_type"]),
                batch_ligand_bond=d(b["batch_ligand_bond"]), time_step=torch.from_numpy(g["time_step"]).to(dev()))


@pytest.mark.parametrize("fixture", ["loss_grad", "loss_grad_ragged"])
def test_diffusion_loss_and_gradients_match_reference(fixture):
    """`loss_grad_ragged`: samples of different sizes in one batch (48 + 8, 40 + 6, 40 + 6, 48 + 8 atoms) -- what the
    reference's training batches are; run as one dense sub-batch per distinct size (training.network_grouped)."""
    g = GU.load(fixture)
    m = _fresh_model()
    m.train()
    kw = _loss_kwargs(g)
    torch.manual_seed(int(g["noise_seed"]))
    res = m.get_diffusion_loss(**kw)
    for k in ("pos", "v", "bond"):
        got, want = float(res["losses"][k]), float(g["loss_" + k])
        print(f"loss {k}: {got:.7g} (reference {want:.7g})")
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)) and abs(got - want) <= 2e-3 * abs(want) + 1e-7
    assert maxabs(res["pred_ligand_pos"], g["out_pred_ligand_pos"]) < 1e-4
    assert maxabs(res["pred_ligand_v"], g["out_pred_ligand_v"]) < 1e-4
    assert maxabs(res["x0"], g["out_x0"]) < 1e-5
    loss = res["losses"]["pos"] + 100.0 * res["losses"]["v"] + 100.0 * res["losses"]["bond"]
    loss.backward()
    params = dict(m.named_parameters())
    worst = 0.0
    for key in [k for k in g.files if k.startswith("grad__")]:
        name = key[len("grad__"):].replace("__", ".")
        want = torch.from_numpy(g[key])
        got = params[name].grad.cpu()
        rel = float((got - want).abs().max() / want.abs().max().clamp(min=1e-12))
        worst = max(worst, rel)
        assert rel < 2e-3, (name, rel)
    names = [str(n) for n in g["grad_norm_names"]]
    got_norms = np.array([float(params[n].grad.double().norm()) if params[n].grad is not None else 0.0 for n in names])
    rel_n = np.abs(got_norms - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-6 * g["grad_norms"].max())
    print(f"{len(names)} parameter gradients: worst relative tensor error {worst:.2g}, worst relative norm error {rel_n.max():.2g}")
    assert rel_n.max() < 2e-3
    assert all(params[n].grad is not None for n in names)


@pytest.mark.parametrize("fixture", ["loss_grad", "loss_grad_ragged"])
def test_validation_loss_uses_the_fused_forward_and_agrees(fixture):
    """torch.no_grad() (the reference's validate()): the network output comes from the fused dd_forward kernels."""
    g = GU.load(fixture)
    m = _fresh_model()
    kw = _loss_kwargs(g)
    torch.manual_seed(int(g["noise_seed"]))
    a = m.get_diffusion_loss(**kw)
    with torch.no_grad():
        torch.manual_seed(int(g["noise_seed"]))
        v = m.get_diffusion_loss(**kw)
    for k in ("pos", "v", "bond"):
        assert abs(float(a["losses"][k]) - float(v["losses"][k])) < 1e-5 * max(1.0, abs(float(a["losses"][k])))
        assert abs(float(v["losses"][k]) - float(g["loss_" + k])) <= 1e-4 * max(1.0, abs(float(g["loss_" + k])))
    assert not v["losses"]["pos"].requires_grad and a["losses"]["pos"].requires_grad
    assert maxabs(a["pred_ligand_pos"], v["pred_ligand_pos"]) < 2e-5


def test_optimizer_steps_reduce_the_loss_and_sampling_sees_the_new_weights():
    g = GU.load("loss_grad")
    m = _fresh_model()
    kw = _loss_kwargs(g)
    opt = torch.optim.Adam(m.parameters(), lr=2e-4)
    losses = []
    for it in range(4):
        torch.manual_seed(5)                                 # same noise every step: the loss must go down
        res = m.get_diffusion_loss(**kw)
        loss = res["losses"]["pos"] + 100.0 * res["losses"]["v"] + 100.0 * res["losses"]["bond"]
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    print("training losses:", " ".join(f"{l:.5f}" for l in losses))
    assert losses[-1] < losses[0]
    m.eval()
    with torch.no_grad():                                    # fused kernels must run on the UPDATED parameters
        torch.manual_seed(5)
        fused = m.get_diffusion_loss(**kw)
    torch.manual_seed(5)
    auto = m.get_diffusion_loss(**kw)
    assert abs(float(fused["losses"]["pos"]) - float(auto["losses"]["pos"])) < 1e-5 * max(1.0, float(auto["losses"]["pos"]))


def test_loss_rejects_layouts_it_would_get_wrong():
    """Unsorted batch vectors / a differently ordered bond list raise in BOTH paths (autograd and fused) instead of giving
    silently wrong triplets and gradients."""
    g = GU.load("loss_grad")
    m = _fresh_model()
    kw = _loss_kwargs(g)
    bad = dict(kw)
    bad["ligand_fc_bond_index"] = kw["ligand_fc_bond_index"].flip(0)            # src-major instead of dst-major
    for grad in (True, False):
        with torch.set_grad_enabled(grad), pytest.raises(NotImplementedError, match="dst-major"):
            m.get_diffusion_loss(**bad)
    bad = dict(kw)
    bl = kw["batch_ligand"].clone()
    bl[0], bl[-1] = bl[-1].item(), bl[0].item()
    bad["batch_ligand"] = bl
    for grad in (True, False):
        with torch.set_grad_enabled(grad), pytest.raises(NotImplementedError, match="sorted"):
            m.get_diffusion_loss(**bad)


def test_in_place_parameter_update_is_seen_by_the_fused_path():
    """Parameters changed in place after a fused call (an optimizer step on the caller's own loss, an EMA copy): the packed
    weight arena is keyed on the parameters' version counters, so the next fused call runs on the new values."""
    g = GU.load("loss_grad")
    m = _fresh_model()
    kw = _loss_kwargs(g)
    with torch.no_grad():
        torch.manual_seed(5)
        a = m.get_diffusion_loss(**kw)
        dict(m.named_parameters())["v_inference.2.weight"].mul_(1.5)     # in place: no train()/load_state_dict in between
        dict(m.named_parameters())["refine_net.base_block.3.lin_node.bias"].add_(0.05)
        torch.manual_seed(5)
        b = m.get_diffusion_loss(**kw)
    assert maxabs(a["pred_ligand_v"], b["pred_ligand_v"]) > 1e-3 and maxabs(a["pred_ligand_pos"], b["pred_ligand_pos"]) > 1e-6
    m2 = _fresh_model()
    with torch.no_grad():
        dict(m2.named_parameters())["v_inference.2.weight"].mul_(1.5)
        dict(m2.named_parameters())["refine_net.base_block.3.lin_node.bias"].add_(0.05)
        torch.manual_seed(5)
        c = m2.get_diffusion_loss(**kw)
    assert torch.equal(b["pred_ligand_v"], c["pred_ligand_v"]) and torch.equal(b["pred_ligand_pos"], c["pred_ligand_pos"])


@pytest.mark.parametrize("rows,out", [(1, 128), (31, 5), (600, 16), (4097, 128), (70001, 64), (70001, 8)])
def test_linear128_forward_and_backward_on_own_gemms(rows, out):
    """training.linear128: y = x W^T + b, dX and dW through dd_gemm128 / dd_gemm128_tn (no ATen GEMM) against torch in
    float64, for the row counts of the step (nodes ... triplets) and every output width the network has (128, 64-wide
    check, 16 coordinate heads, 8 / 5 class heads), incl. a row count that is no multiple of the 32-row trips."""
    from decompdiff_amd import training
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + out)
    x = torch.randn(rows, 128, generator=g).to(dev).requires_grad_(True)
    W = (torch.randn(out, 128, generator=g) * 0.1).to(dev).requires_grad_(True)
    b = torch.randn(out, generator=g).to(dev).requires_grad_(True)
    dy = torch.randn(rows, out, generator=g).to(dev)
    y = training.linear128(x, W, b)
    y.backward(dy)
    xd, Wd, bd = (t.detach().double() for t in (x, W, b))
    want_y = xd @ Wd.t() + bd
    want_dx, want_dW, want_db = dy.double() @ Wd, dy.double().t() @ xd, dy.double().sum(0)
    rel = lambda a, w: float((a.double() - w).abs().max() / w.abs().max().clamp(min=1e-30))
    errs = dict(y=rel(y.detach(), want_y), dx=rel(x.grad, want_dx), dW=rel(W.grad, want_dW), db=rel(b.grad, want_db))
    print(f"linear128 rows={rows} out={out}:", {k: f"{v:.2g}" for k, v in errs.items()})
    assert max(errs.values()) < 2e-6
    # reproducible bit for bit (fixed slab order, no atomics)
    x2, W2 = x.detach().clone().requires_grad_(True), W.detach().clone().requires_grad_(True)
    training.linear128(x2, W2, b.detach()).backward(dy)
    assert torch.equal(W2.grad, W.grad) and torch.equal(x2.grad, x.grad)


@pytest.mark.parametrize("rows,kf", [(33, 13), (5000, 20), (42240, 84), (97440, 13)])
def test_linear_feat_backward_on_own_gemms(rows, kf):
    """training.linear_feat (narrow feature block -> 128 hidden channels): forward = ATen, backward through dd_gemm128
    (df = dY W) and dd_gemm128_tn (dW^T = f^T dY) against torch in float64."""
    from decompdiff_amd import training
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + kf)
    f = torch.rand(rows, kf, generator=g).to(dev).requires_grad_(True)
    W = (torch.randn(128, kf, generator=g) * 0.1).to(dev).requires_grad_(True)
    dy = torch.randn(rows, 128, generator=g).to(dev)
    training.linear_feat(f, W).backward(dy)
    want_df, want_dW = dy.double() @ W.detach().double(), dy.double().t() @ f.detach().double()
    rel = lambda a, w: float((a.double() - w).abs().max() / w.abs().max().clamp(min=1e-30))
    errs = dict(df=rel(f.grad, want_df), dW=rel(W.grad, want_dW))
    print(f"linear_feat rows={rows} kf={kf}:", {k: f"{v:.2g}" for k, v in errs.items()})
    assert max(errs.values()) < 2e-6


def test_graphed_train_step_follows_the_eager_steps():
    """training.GraphedTrainStep: get_diffusion_loss + backward + Adam as one captured graph per batch shape.  Same seeds, same
    batches: the captured iterations must follow the eager ones (same losses step by step, same parameters at the end up to the
    re-association of atomic gradient sums), and a batch of another shape gets its own graph."""
    from decompdiff_amd import training
    torch.manual_seed(3)
    b1 = synth.build_sampling_batch(synth.make_pocket(31, 80, (4, 4), 6, num_full_protein=0), 3)
    b2 = synth.build_sampling_batch(synth.make_pocket(32, 64, (3, 3), 5, num_full_protein=0), 2)
    d = lambda t: t.to(dev()) if torch.is_tensor(t) else t
    to_kw = lambda b: dict(
        protein_pos=d(b["protein_pos"]), protein_v=d(b["protein_v"]), batch_protein=d(b["batch_protein"]),
        protein_group_idx=d(b["protein_group_idx"]), ligand_pos=d(b["init_ligand_pos"]), ligand_v=d(b["init_ligand_v"]),
        ligand_v_aux=d(b["ligand_v_aux"]), batch_ligand=d(b["batch_ligand"]), ligand_group_idx=d(b["ligand_group_idx"]),
        prior_centers=d(b["prior_centers"]), prior_stds=d(b["prior_stds"]), prior_num_atoms=d(b["prior_num_atoms"]),
        batch_prior=d(b["batch_prior"]), prior_group_idx=d(b["prior_group_idx"]), ligand_decomp_batch=d(b["ligand_decomp_batch"]),
        ligand_decomp_index=d(b["ligand_decomp_index"]), ligand_fc_bond_index=d(b["ligand_fc_bond_index"]),
        ligand_fc_bond_type=d(b["init_ligand_fc_bond_type"]), batch_ligand_bond=d(b["batch_ligand_bond"]))
    kws = [to_kw(b1), to_kw(b2)]

    def fresh(capturable):
        m = DecompScorePosNet3D(shipped_config(), 29, 10, 8)
        sd = m.state_dict(); sd.update(synth.synthetic_state_dict(shipped_config(), 1)); m.load_state_dict(sd)
        m = m.to(dev()).train()
        return m, torch.optim.Adam(m.parameters(), lr=1e-4, capturable=capturable)

    order = [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 0, 1, 0]
    m_e, opt_e = fresh(False)
    torch.manual_seed(11)
    eager = []
    for i in order:
        opt_e.zero_grad(set_to_none=True)
        r = m_e.get_diffusion_loss(**kws[i])
        loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
        loss.backward(); opt_e.step()
        eager.append(float(loss))
    m_g, opt_g = fresh(True)
    gs = training.GraphedTrainStep(m_g, opt_g, loss_weights=(1.0, 100.0, 100.0), warmup=2)
    torch.manual_seed(11)
    graphed = [float(gs.step(**kws[i])["loss"]) for i in order]
    assert gs.replays == len(order) - 4 and gs.eager_steps == 4 and len(gs._graphs) == 2
    for a, c in zip(eager, graphed):
        assert abs(a - c) <= 2e-4 * max(1.0, abs(a)), (eager, graphed)
    # (Adam normalises every gradient component by its own running magnitude: components at the noise level of the atomic
    #  gradient sums move by +-lr per step in either run -- the parameters are held to a fraction of the distance they can travel)
    worst = max(float((p - q).abs().max()) for p, q in zip(m_e.parameters(), m_g.parameters()))
    assert worst < 0.5 * len(order) * 1e-4, worst
    with pytest.raises(ValueError):
        training.GraphedTrainStep(m_e, opt_e)


def test_graph_entries_pin_the_cached_structures_they_captured():
    """a captured step bakes in the device addresses of the cached index structures / segment plans it read.  The
    module-level caches evict (16 structures, 64 plans); the graph entry must keep what its capture used alive.  Capture a graph, churn
    more than 16 other structures and more than 64 plans, replay: the replayed steps must follow an eager twin step for step -- and with
    gradient clipping (max_grad_norm, the reference's train loop) in both."""
    from decompdiff_amd import training
    torch.manual_seed(5)
    b1 = synth.build_sampling_batch(synth.make_pocket(41, 72, (4, 3), 5, num_full_protein=0), 2)
    d = lambda t: t.to(dev()) if torch.is_tensor(t) else t
    kw = dict(
        protein_pos=d(b1["protein_pos"]), protein_v=d(b1["protein_v"]), batch_protein=d(b1["batch_protein"]),
        protein_group_idx=d(b1["protein_group_idx"]), ligand_pos=d(b1["init_ligand_pos"]), ligand_v=d(b1["init_ligand_v"]),
        ligand_v_aux=d(b1["ligand_v_aux"]), batch_ligand=d(b1["batch_ligand"]), ligand_group_idx=d(b1["ligand_group_idx"]),
        prior_centers=d(b1["prior_centers"]), prior_stds=d(b1["prior_stds"]), prior_num_atoms=d(b1["prior_num_atoms"]),
        batch_prior=d(b1["batch_prior"]), prior_group_idx=d(b1["prior_group_idx"]), ligand_decomp_batch=d(b1["ligand_decomp_batch"]),
        ligand_decomp_index=d(b1["ligand_decomp_index"]), ligand_fc_bond_index=d(b1["ligand_fc_bond_index"]),
        ligand_fc_bond_type=d(b1["init_ligand_fc_bond_type"]), batch_ligand_bond=d(b1["batch_ligand_bond"]))

    def fresh(capturable):
        m = DecompScorePosNet3D(shipped_config(), 29, 10, 8)
        sd = m.state_dict(); sd.update(synth.synthetic_state_dict(shipped_config(), 2)); m.load_state_dict(sd)
        m = m.to(dev()).train()
        return m, torch.optim.Adam(m.parameters(), lr=1e-4, capturable=capturable)

    n_steps = 7
    m_e, opt_e = fresh(False)
    torch.manual_seed(21)
    eager = []
    for _ in range(n_steps):
        opt_e.zero_grad(set_to_none=True)
        r = m_e.get_diffusion_loss(**kw)
        loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m_e.parameters(), 8.0)
        opt_e.step()
        eager.append(float(loss))
    m_g, opt_g = fresh(True)
    gs = training.GraphedTrainStep(m_g, opt_g, loss_weights=(1.0, 100.0, 100.0), warmup=2, max_grad_norm=8.0)
    torch.manual_seed(21)
    graphed = [float(gs.step(**kw)["loss"]) for _ in range(3)]            # 2 eager + the capture's replay
    ent = next(iter(gs._graphs.values()))
    assert len(ent["keep"]) > 0                                            # the structures / plans the capture read
    held = {id(o) for o in ent["keep"]}
    # churn the caches past their limits
    for i in range(training._STRUCT_MAX_ENTRIES + 3):
        training._structure(1, 40 + i, 6, 8, dev())
    for i in range(70):
        nl = [5 + (i % 7), 6 + (i // 7)]
        training._static_plan(torch.arange(2, device=dev()).repeat_interleave(torch.tensor(nl, device=dev())), 2, nl, False)
    assert not any(id(v) in held for v in training._STRUCT.values())       # evicted from the cache ...
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), -7, dtype=torch.int64, device=dev()) for _ in range(8)]   # ... and their memory would be reused by now
    graphed += [float(gs.step(**kw)["loss"]) for _ in range(n_steps - 3)]
    del junk
    assert gs.replays == n_steps - 2
    for a, c in zip(eager, graphed):
        assert np.isfinite(c) and abs(a - c) <= 2e-4 * max(1.0, abs(a)), (eager, graphed)


def test_gemm128_tn_bias_matches_torch():
    """dd_gemm128_tn_bias: dW = dY^T X and db = column sums of dY from one launch pair, against torch in float64, for the three
    tile heights (M <= 32 / 64 / 128) and row counts that are not multiples of the 32-row trips."""
    import ctypes
    from decompdiff_amd import hip_lib
    lib = hip_lib.load()
    torch.manual_seed(2)
    for rows, M in ((1000, 128), (4097, 128), (333, 16), (70000, 64), (129, 5)):
        dy = torch.randn(rows, M, device=dev())
        x = torch.randn(rows, 128, device=dev())
        dW = torch.empty(M, 128, device=dev()); db = torch.empty(M, device=dev())
        scratch = torch.empty(int(lib.dd_gemm128_tn_scratch_floats(rows, M)), device=dev())
        hip_lib.check(lib.dd_gemm128_tn_bias(hip_lib.ptr(dy), M, M, hip_lib.ptr(x), 128, rows, hip_lib.ptr(scratch), hip_lib.ptr(dW), 128, 0,
                                             hip_lib.ptr(db), hip_lib.stream_ptr(dev())), "dd_gemm128_tn_bias")
        want_W = (dy.double().t() @ x.double()); want_b = dy.double().sum(0)
        assert float((dW.double() - want_W).abs().max()) < 2e-4 * max(1.0, float(want_W.abs().max())), (rows, M)
        assert float((db.double() - want_b).abs().max()) < 2e-4 * max(1.0, float(want_b.abs().max())), (rows, M)
        dW2 = torch.empty_like(dW)
        hip_lib.check(lib.dd_gemm128_tn(hip_lib.ptr(dy), M, M, hip_lib.ptr(x), 128, rows, hip_lib.ptr(scratch), hip_lib.ptr(dW2), 128, 0,
                                        hip_lib.stream_ptr(dev())), "dd_gemm128_tn")
        assert torch.equal(dW, dW2)                           # the weight gradient does not depend on the bias path


def test_fused_layernorm_relu_matches_torch():
    """training.ln_relu (dd_ln_relu_forward / _backward): y, dx, dgamma, dbeta against torch autograd in float64, row counts from
    one row to more rows than the backward has partial-sum slabs."""
    from decompdiff_amd import training
    torch.manual_seed(4)
    for rows in (1, 3, 130, 4099, 97440):
        x = torch.randn(rows, 128, device=dev()) * 1.7 + 0.3
        g = torch.randn(128, device=dev()) * 0.5 + 1.0
        b = torch.randn(128, device=dev()) * 0.3
        dy = torch.randn(rows, 128, device=dev())
        xs, gs, bs = (t.clone().requires_grad_(True) for t in (x, g, b))
        y = training.ln_relu(xs, gs, bs)
        y.backward(dy)
        xd, gd, bd = x.double(), g.double(), b.double()
        yd = torch.relu(torch.nn.functional.layer_norm(xd, (128,), gd, bd, 1e-5))
        assert float((y.double() - yd).abs().max()) < 1e-5
        # the backward in float64 WITH THE KERNEL'S OWN ReLU mask (y > 0): a pre-activation within rounding of 0 may take either
        # side in fp32, and one such element moves a whole row's dx -- the mask is part of the forward result, not of the test
        mask = (y > 0).double()
        mean = xd.mean(1, keepdim=True)
        rstd = 1.0 / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5)
        xhat = (xd - mean) * rstd
        dz = dy.double() * mask
        dxhat = dz * gd
        want_dx = rstd * (dxhat - dxhat.mean(1, keepdim=True) - xhat * (dxhat * xhat).mean(1, keepdim=True))
        assert float((xs.grad.double() - want_dx).abs().max()) < 2e-5 * max(1.0, float(want_dx.abs().max())), rows
        for a, w in ((gs.grad, (dz * xhat).sum(0)), (bs.grad, dz.sum(0))):
            assert float((a.double() - w).abs().max()) < 1e-4 * max(1.0, float(w.abs().max())), rows
        y2 = training.ln_relu(xs, gs, bs)
        assert torch.equal(y, y2)                              # run to run bit-identical (no atomics)


def _grads_agree(ga, gb, g=None):
    """Two sets of parameter gradients of one batch agree: per tensor in the 2-norm (a training backward is not bit-reproducible --
    ATen's atomic index_add_ orders differ from run to run, and a tensor with a small gradient then moves by ~1 % of its largest
    element in a few entries between two runs of the SAME code), and, if the reference's fixture is given, against its stored
    tensors / norms with the bounds of test_diffusion_loss_and_gradients_match_reference."""
    assert set(ga) == set(gb)
    big = max(float(v.double().norm()) for v in gb.values())
    for n in gb:
        den = max(float(gb[n].double().norm()), 1e-3 * big)
        assert float((ga[n].double() - gb[n].double()).norm()) / den < 5e-3, n
    assert all(bool(torch.isfinite(v).all()) for v in ga.values())
    if g is not None:
        for key in [k for k in g.files if k.startswith("grad__")]:
            name = key[len("grad__"):].replace("__", ".")
            want = torch.from_numpy(g[key])
            assert float((ga[name].cpu() - want).abs().max() / want.abs().max().clamp(min=1e-12)) < 2e-3, name
        names = [str(n) for n in g["grad_norm_names"]]
        got = np.array([float(ga[n].double().norm()) if n in ga else 0.0 for n in names])
        assert (np.abs(got - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-6 * g["grad_norms"].max())).max() < 2e-3


def test_padded_heterogeneous_training_batch_equals_size_groups(monkeypatch):
    """training.network_padded: a batch of different complexes as ONE padded dense pass (padding atoms out of the kNN graph, masked
    in the bond-graph / triplet attentions) against one dense pass per distinct size (DD_TRAIN_PAD=0): same losses, same gradients
    up to the association of the row sums -- and both against the reference's own numbers (`loss_grad_ragged`)."""
    g = GU.load("loss_grad_ragged")
    kw = _loss_kwargs(g)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DD_TRAIN_PAD", mode)
        m = _fresh_model(); m.train()
        torch.manual_seed(int(g["noise_seed"]))
        r = m.get_diffusion_loss(**kw)
        loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
        loss.backward()
        res[mode] = (r, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    (ra, ga), (rb, gb) = res["1"], res["0"]
    for k in ("pos", "v", "bond"):
        assert abs(float(ra["losses"][k]) - float(rb["losses"][k])) <= 2e-6 * max(1.0, abs(float(rb["losses"][k]))), k
        assert abs(float(ra["losses"][k]) - float(g["loss_" + k])) <= 1e-4 * max(1.0, abs(float(g["loss_" + k])))
    assert maxabs(ra["pred_ligand_pos"], rb["pred_ligand_pos"]) < 1e-5 and maxabs(ra["pred_ligand_v"], rb["pred_ligand_v"]) < 1e-5
    _grads_agree(ga, gb, g)


@pytest.mark.parametrize("bucket", [(1, 1), (32, 4)])
def test_padded_objective_equals_the_ragged_objective(bucket):
    """training.objective_padded on pad_prepared(...) -- fixed shapes, per-sample means over the real rows -- against
    training.objective on the same prepared ragged batch: same losses, same gradients (also with sizes rounded up to a bucket,
    i.e. whole padding atoms beyond the batch's largest sample), and the reference's own losses (`loss_grad_ragged`)."""
    from decompdiff_amd import training
    g = GU.load("loss_grad_ragged")
    kw = _loss_kwargs(g)
    names = ("protein_pos", "protein_v", "batch_protein", "ligand_pos", "ligand_v", "ligand_v_aux", "batch_ligand", "prior_centers",
             "prior_stds", "prior_num_atoms", "batch_prior", "ligand_decomp_batch", "ligand_fc_bond_index", "ligand_fc_bond_type",
             "batch_ligand_bond")
    out = {}
    for mode in ("ragged", "padded"):
        m = _fresh_model(); m.train()
        torch.manual_seed(int(g["noise_seed"]))
        prep = training.prepare_batch(m, *[kw[n] for n in names], time_step=kw["time_step"])
        if mode == "ragged":
            r = training.objective(m, prep)
        else:
            pp = training.pad_prepared(m, prep, bucket)
            assert pp is not None and pp["NPm"] % bucket[0] == 0 and pp["NLm"] % bucket[1] == 0
            r = training.objective_padded(m, pp)
        loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
        loss.backward()
        out[mode] = (r, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    (ra, ga), (rb, gb) = out["padded"], out["ragged"]
    for k in ("pos", "v", "bond"):
        assert abs(float(ra["losses"][k]) - float(rb["losses"][k])) <= 2e-6 * max(1.0, abs(float(rb["losses"][k]))), k
        assert abs(float(ra["losses"][k]) - float(g["loss_" + k])) <= 1e-4 * max(1.0, abs(float(g["loss_" + k])))
    _grads_agree(ga, gb, g)


def test_graphed_train_step_with_batches_of_different_complexes():
    """GraphedTrainStep on mixed-size batches: batches whose largest protein / ligand fall into one shape bucket share ONE captured
    graph (padded layout); the captured iterations follow the eager ones."""
    from decompdiff_amd import training
    torch.manual_seed(8)
    mk = lambda seed, shapes: synth.concat_sampling_batches(
        [synth.build_sampling_batch(synth.make_pocket(seed + i, n_p, arms, sca, num_full_protein=0), 1) for i, (n_p, arms, sca) in enumerate(shapes)])
    batches = [mk(60, [(70, (3, 3), 4), (64, (2, 3), 3), (58, (3, 2), 5)]),          # largest: 70 + 10  -> bucket (96, 12)
               mk(70, [(66, (3, 3), 5), (72, (3, 3), 3), (60, (2, 2), 4)]),          # largest: 72 + 11  -> the same bucket
               mk(80, [(90, (4, 3), 4), (65, (2, 2), 5), (69, (3, 3), 6)])]          # largest: 90 + 12  -> the same bucket
    d = lambda t: t.to(dev()) if torch.is_tensor(t) else t
    to_kw = lambda b: dict(
        protein_pos=d(b["protein_pos"]), protein_v=d(b["protein_v"]), batch_protein=d(b["batch_protein"]),
        protein_group_idx=d(b["protein_group_idx"]), ligand_pos=d(b["init_ligand_pos"]), ligand_v=d(b["init_ligand_v"]),
        ligand_v_aux=d(b["ligand_v_aux"]), batch_ligand=d(b["batch_ligand"]), ligand_group_idx=d(b["ligand_group_idx"]),
        prior_centers=d(b["prior_centers"]), prior_stds=d(b["prior_stds"]), prior_num_atoms=d(b["prior_num_atoms"]),
        batch_prior=d(b["batch_prior"]), prior_group_idx=d(b["prior_group_idx"]), ligand_decomp_batch=d(b["ligand_decomp_batch"]),
        ligand_decomp_index=d(b["ligand_decomp_index"]), ligand_fc_bond_index=d(b["ligand_fc_bond_index"]),
        ligand_fc_bond_type=d(b["init_ligand_fc_bond_type"]), batch_ligand_bond=d(b["batch_ligand_bond"]))
    kws = [to_kw(b) for b in batches]

    def fresh(capturable):
        m = DecompScorePosNet3D(shipped_config(), 29, 10, 8)
        sd = m.state_dict(); sd.update(synth.synthetic_state_dict(shipped_config(), 1)); m.load_state_dict(sd)
        m = m.to(dev()).train()
        return m, torch.optim.Adam(m.parameters(), lr=1e-4, capturable=capturable)

    order = [0, 1, 2, 0, 1, 2, 1, 0]
    m_e, opt_e = fresh(False)
    torch.manual_seed(12)
    eager = []
    for i in order:
        opt_e.zero_grad(set_to_none=True)
        r = m_e.get_diffusion_loss(**kws[i])
        loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
        loss.backward(); opt_e.step()
        eager.append(float(loss))
    m_g, opt_g = fresh(True)
    gs = training.GraphedTrainStep(m_g, opt_g, loss_weights=(1.0, 100.0, 100.0), warmup=2, bucket=(32, 4))
    torch.manual_seed(12)
    graphed = [float(gs.step(**kws[i])["loss"]) for i in order]
    assert len(gs._graphs) == 1 and gs.replays == len(order) - 2 and gs.eager_steps == 2      # one bucket, one graph
    for a, c in zip(eager, graphed):
        assert abs(a - c) <= 2e-4 * max(1.0, abs(a)), (eager, graphed)

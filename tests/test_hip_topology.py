"""GPU: the topologies workflow.train_exp exposes beyond its defaults (hidden encoder /
decoder layers, dropout, the per-subject output scale head; reference networks.py:16-20,
51-59,66-77) on the HIP path's general chain of launches (csrc/mopoe_general.inc).

The training steps of every `t_*` fixture recorded from the reference run in
test_hip_parity.py (oracle's eps and dropout keep masks injected).  Here: evaluation
forwards (Dropout = identity) with the flags of BaseMMVae.forward, the device-drawn masks,
the mirror modules' surface, the zero_grad / backward / step sequence, the N-rank step."""
from collections import OrderedDict
from importlib import import_module

import pytest
import torch

import mopoe_amd as mm
import mopoe_oracle as mo
from hip_util import Report, TOL, compare_forward, make_engine
from surface_util import make_experiment, run_epochs

pytestmark = pytest.mark.gpu
L = mm._lib
C1 = dict(names=["clinical", "rois"], input_dim=[7, 444], style_dim=[3, 20])
C5 = dict(names=["clinical", "rois", "snps", "tracts"], input_dim=[7, 444, 128, 64], style_dim=[3, 3, 3, 3])
TOPOS = [
    dict(enc_layers=0, dec_layers=0),      # (the row-group kernel on x: no chain of launches)
    dict(enc_layers=2, dec_layers=1),
    dict(enc_layers=0, dec_layers=0, sample_scale=True),
    dict(enc_layers=3, dec_layers=2, dropout=0.3, sample_scale=True),
    dict(enc_layers=1, dec_layers=0, dropout=0.1),
]


@pytest.mark.parametrize("topo", TOPOS, ids=lambda t: "-".join("%s%s" % (k[:3], v) for k, v in t.items()))
@pytest.mark.parametrize("method,base", [("joint_elbo", C1), ("poe", C1), ("moe", C1), ("joint_elbo", C5)])
def test_evaluation_forwards_match_oracle(topo, method, base):
    cfg = mo.Config(method=method, **base, **topo)
    spec, eng = make_engine(cfg)
    assert spec.general
    params = mo.init_params(cfg, 0)
    present = cfg.names if base is C1 else ["rois", "snps", "tracts"]
    rep = Report("fwd %s %s" % (method, topo))
    for n, sample, expert in ((37, True, None), (16, False, None), (21, True, "_".join(sorted(present))),
                              (2304, False, None)):
        x = mo.make_inputs(cfg.names, cfg.input_dim, n, seed=n, present=present)
        noise = mo.Noise(generator=mo.noise_rng(n))
        with torch.no_grad():
            out = mo.forward(params, cfg, x, noise, sample_latents=sample, use_expert=expert)
        plan, ws = eng.forward(x, sample=sample, use_expert=expert, eps=noise.tape)
        torch.cuda.synchronize()
        compare_forward(rep, spec, eng, plan, ws, out, prefix="n%d/" % n)
    rep.finish()


def test_device_drawn_dropout_masks():
    """Without injected masks the kernels draw them (Philox, keyed by seed / step / layer /
    element): a unit is kept with probability 1 - p and scaled by 1 / (1 - p); the same step
    of a twin engine draws the same masks, the next step different ones."""
    cfg = mo.Config(**C1, enc_layers=2, dec_layers=1, dropout=0.25)
    _, a = make_engine(cfg)
    _, b = make_engine(cfg)
    x = mo.make_inputs(cfg.names, cfg.input_dim, 512, seed=1)
    eps = mo.Noise(generator=mo.noise_rng(2))
    mo.forward(mo.init_params(cfg, 0), cfg, x, eps, train=False)
    kept = []
    for step in range(2):
        w0 = {n: (a.views["encoders.%s.shared_encoder.0.weight" % n].clone(),
                  a.views["encoders.%s.shared_encoder.0.bias" % n].clone()) for n in cfg.names}
        (_, wa), (_, wb) = a.train_step(x, eps=eps.tape), b.train_step(x, eps=eps.tape)
        torch.cuda.synchronize()
        for m, name in enumerate(cfg.names):
            # (Dropout sits in the layer's own epilogue: the pre-dropout activation is not kept,
            #  so it is recomputed here from the weights the step read)
            pre = torch.relu(x[name].cuda().double() @ w0[name][0].double().t() + w0[name][1].double()).float()
            act = wa.enc_act[m][0]
            live = pre > 1e-4
            keep = (act != 0)[live].float().mean().item()
            assert abs(keep - 0.75) < 0.01, keep
            on = live & (act != 0)
            assert torch.allclose(act[on], pre[on] * (1.0 / 0.75), rtol=1e-4, atol=1e-6)
            assert not (act[pre == 0] != 0).any()
            assert torch.equal(wa.enc_act[m][1], wb.enc_act[m][1])     # twin: the same masks
            kept.append((act != 0).clone())
        assert torch.equal(a.params, b.params)
    assert not torch.equal(kept[0], kept[2])        # another step, other masks
    a.check_valid(sync=True)
    loss = float(wa.stats[L.STAT_TOTAL_LOSS])
    assert loss == loss and abs(loss) < 1e7


@pytest.mark.parametrize("method", ["joint_elbo", "poe"])
def test_mirror_modules_and_the_train_sequence(method):
    """The mirror Encoder / Decoder carry the reference's state_dict keys for a general
    topology; zero_grad / backward / step (run_epochs.py:180-182) in evaluation-free
    training mode with dropout 0 equals the oracle."""
    cfg = mo.Config(**C1, method=method, enc_layers=2, dec_layers=2, sample_scale=True)
    exp = make_experiment(cfg, "cuda")
    model = exp.models
    assert list(model.state_dict().keys()) == list(mo.param_shapes(cfg).keys())
    params = mo.init_params(cfg, 0)
    model.load_state_dict(params)
    state = mo.adam_init(params)
    x = mo.make_inputs(cfg.names, cfg.input_dim, 48, seed=50)
    # standalone module forwards (mopoe_linear per layer)
    rep = Report("topology surface " + method)
    model.eval()
    for m, name in enumerate(cfg.names):
        got = model.encoders[name](x[name].cuda())
        want = mo.encoder_forward(params, cfg, m, x[name])[:4]
        for g, w, nm in zip(got, want, ("s_mu", "s_lv", "c_mu", "c_lv")):
            rep.close("%s/%s" % (name, nm), g, w, *TOL["latent"])
        zs, zc = torch.randn(48, cfg.style_dim[m]), torch.randn(48, cfg.class_dim)
        loc, scale = model.decoders[name](zs.cuda(), zc.cuda())
        wloc, wscale = mo.decoder_forward(params, cfg, m, zs, zc)
        rep.close(name + "/loc", loc, wloc, *TOL["loc"])
        rep.close(name + "/scale", scale, wscale, 2e-5, 1e-6)
    gen = model.generate(16)
    assert gen["rois"].shape == (16, 444)
    model.train()
    noise = mo.Noise(generator=mo.noise_rng(70))
    out, grads = mo.loss_and_grads(params, cfg, x, noise)
    eng = model.engine
    orig = eng.train_step
    eng.train_step = lambda b, eps=None, **kw: orig(b, eps=noise.tape, **kw)
    batch = (OrderedDict((k, v.double()) for k, v in x.items()), None, {})
    res = run_epochs.basic_routine_epoch(exp, 0, batch)
    eng.train_step = orig
    exp.optimizers.zero_grad()
    res["total_loss"].backward()
    rep.close("total_loss", res["total_loss"], out["total_loss"], *TOL["scalar"])
    named = dict(model.named_parameters())
    for k, g in grads.items():
        assert named[k].grad is not None, k
        rep.close_scaled("grad/" + k, named[k].grad, g, TOL["grad"])
    exp.optimizers.step()
    mo.adam_step(cfg, params, grads, state)
    torch.cuda.synchronize()
    for k, g in grads.items():
        mask = g.abs() > 1e-6
        rep.close("param/" + k, named[k].detach().cpu()[mask], params[k][mask], *TOL["param1"])
    rep.finish()


def test_one_call_rccl_step_of_a_general_topology():
    """mopoe_general_train_step with a communicator (gradients, ncclAllReduce, the Adam
    launch over the topology's segments) leaves the bits of the one-rank step."""
    import socket
    import torch.distributed as dist
    parallel = import_module("2022_cambroise_interpret_multivae_amd.parallel")
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0,
                                world_size=1, device_id=torch.device("cuda", 0))
    cfg = mo.Config(**C1, method="poe", enc_layers=2, dec_layers=1, dropout=0.2)
    _, a = make_engine(cfg)
    _, b = make_engine(cfg)
    one = parallel.DataParallelStep(a, exchange="rccl")
    params = mo.init_params(cfg, 0)
    for step in range(3):
        x = mo.make_inputs(cfg.names, cfg.input_dim, 64, seed=step)
        noise = mo.Noise(generator=mo.noise_rng(step), mask_generator=mo.noise_rng(9 + step))
        mo.loss_and_grads(params, cfg, x, noise)
        one(x, eps=noise.tape, masks=noise.mask_tape)
        b.train_step(x, eps=noise.tape, masks=noise.mask_tape)
    torch.cuda.synchronize()
    a.check_valid(sync=True)
    for name in ("params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert a.adam_steps() == b.adam_steps() == OrderedDict(clinical=3, rois=3)
    one.rccl.close()


@pytest.mark.parametrize("method,base", [("joint_elbo", C1), ("poe", C1), ("joint_elbo", C5)])
def test_epilogue_forms_of_the_chain_equal_the_separate_launches(method, base, monkeypatch):
    """The chain's second cut moved Dropout into the hidden layers'
    epilogues, the likelihood into the output layer's, and split the backward-data GEMM's K over
    a block's waves.  Same seed, same batches, device-drawn noise and masks: Dropout apart
    (MOPOE_DROPOUT_APART) is the same arithmetic on the same Philox draws -- bit-identical
    gradients and scalars; the likelihood apart / one wave per backward-data tile
    (MOPOE_NLL_APART, MOPOE_NN_WIDE) differ in summation order only."""
    import mopoe_amd as mm
    cfg = mo.Config(method=method, **base, enc_layers=2, dec_layers=1, dropout=0.2)
    xs = [mo.make_inputs(cfg.names, cfg.input_dim, 208, seed=70 + i) for i in range(4)]

    def run(**env):
        for k in ("MOPOE_DROPOUT_APART", "MOPOE_NLL_APART", "MOPOE_NN_WIDE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        spec, _ = make_engine(cfg)
        eng = mm.MoPoEEngine(spec, "cuda", seed=4242)
        eng.load_params(mo.init_params(cfg, 0))
        out = []
        for x in xs:     # (gradients only: Adam turns the rounding noise of a ~0 gradient into a step)
            plan, ws = eng.train_step(x, apply_adam=False)
            torch.cuda.synchronize()
            out.append((eng.grads.clone(), ws.stats[:L.NUM_STATS].clone()))
        eng.check_valid(sync=True)
        return out

    base_run = run()
    for (g0, s0), (g1, s1) in zip(base_run, run(MOPOE_DROPOUT_APART="1")):
        assert torch.equal(g0, g1) and torch.equal(s0, s1)
    for env in (dict(MOPOE_NLL_APART="1"), dict(MOPOE_NN_WIDE="1")):
        for (g0, s0), (g2, s2) in zip(base_run, run(**env)):
            assert (g0 - g2).abs().max().item() <= 2e-5 * g0.abs().max().item() + 1e-7, env
            assert torch.allclose(s0, s2, rtol=2e-5, atol=1e-4), env


RANDOM_CHAINS = [
    # names / input dims / style dims / class dim / method / topology / rows: shapes off the
    # benchmark's -- z wider than the 48 the fused decoder layer takes, dims that are not multiples
    # of 4 or 16, one to five modalities, a missing one, partial row tiles
    dict(dims=[13, 70], style=[0, 30], D=24, method="joint_elbo", topo=dict(enc_layers=2, dec_layers=2, dropout=0.3), n=37),
    dict(dims=[5, 33, 130], style=[2, 9, 28], D=20, method="joint_elbo", topo=dict(enc_layers=1, dec_layers=1, dropout=0.1), n=200),
    dict(dims=[444], style=[7], D=16, method="joint_elbo", topo=dict(enc_layers=3, dec_layers=1), n=64),
    dict(dims=[9, 18, 27, 36, 45], style=[2, 2, 3, 4, 5], D=11, method="joint_elbo",
         topo=dict(enc_layers=2, dec_layers=1, dropout=0.2), n=48, present=[0, 2, 3]),
    dict(dims=[21, 260], style=[4, 12], D=20, method="poe", topo=dict(enc_layers=2, dec_layers=1, dropout=0.25), n=96),
    dict(dims=[21, 260], style=[4, 12], D=20, method="moe", topo=dict(enc_layers=1, dec_layers=2), n=50),
    dict(dims=[30, 17], style=[40, 3], D=20, method="joint_elbo", topo=dict(enc_layers=1, dec_layers=1, dropout=0.2), n=80),
    # the logvar head (the likelihood on ITS launch): two decoder passes per modality, tile-aligned; one pass, ragged
    dict(dims=[12, 100], style=[2, 6], D=10, method="poe", topo=dict(enc_layers=1, dec_layers=1, sample_scale=True), n=32),
    dict(dims=[12, 100, 33], style=[2, 6, 0], D=10, method="joint_elbo",
         topo=dict(enc_layers=2, dec_layers=2, dropout=0.2, sample_scale=True), n=45),
]


@pytest.mark.parametrize("case", RANDOM_CHAINS, ids=lambda c: "%s-%s-n%d" % (c["method"], "x".join(map(str, c["dims"])), c["n"]))
def test_chains_of_other_shapes_against_the_oracle(case):
    """Two training steps of general topologies whose shapes are not the benchmark's, the oracle's
    eps and dropout keep masks injected, step by step from the engine's own parameters: forward
    scalars and reconstructions, every gradient, Adam's first moment."""
    names = ["m%d" % i for i in range(len(case["dims"]))]
    cfg = mo.Config(names=names, input_dim=case["dims"], style_dim=case["style"], class_dim=case["D"],
                    method=case["method"], **case["topo"])
    present = None if "present" not in case else [names[i] for i in case["present"]]
    spec, eng = make_engine(cfg)
    assert spec.general
    rep = Report("chain %s" % case)
    for step in range(2):
        x = mo.make_inputs(cfg.names, cfg.input_dim, case["n"], seed=300 + step, present=present)
        noise = mo.Noise(generator=mo.noise_rng(310 + step), mask_generator=mo.noise_rng(320 + step))
        params_now = OrderedDict((k, v.cpu().clone()) for k, v in eng.named_params().items())
        st = {"step": OrderedDict((k, step if present is None or k.split(".")[1] in present else 0)
                                  for k in params_now),
              "exp_avg": OrderedDict((k, v.cpu().clone()) for k, v in spec.param_views(eng.exp_avg).items()),
              "exp_avg_sq": OrderedDict((k, v.cpu().clone()) for k, v in spec.param_views(eng.exp_avg_sq).items())}
        out, grads = mo.train_step(params_now, cfg, x, noise, st)
        plan, ws = eng.train_step(x, eps=noise.tape, masks=noise.mask_tape or None)
        torch.cuda.synchronize()
        p = "step%d/" % step
        compare_forward(rep, spec, eng, plan, ws, out, prefix=p, check_scale=False)
        for k, g in grads.items():
            rep.close_scaled(p + "grad/" + k, eng.grad_views[k], g, TOL["grad"])
            rep.close_scaled(p + "exp_avg/" + k, spec.param_views(eng.exp_avg)[k], st["exp_avg"][k], TOL["grad"])
        if cfg.sample_scale:    # the head's (N, d) scale of THIS forward
            res = eng.results(plan, ws)
            for k in res["rec"]:
                rep.close(p + "rec/%s/scale" % k, res["rec"][k].scale, out["results"]["rec"][k][1], 2e-5, 1e-6)
        eng.check_valid(sync=True)
    rep.finish()


def _kernel_counts(eng, batches, **kw):
    L.profile_enable(True)
    try:
        for x in batches:
            eng.train_step(x, **kw)
        torch.cuda.synchronize()
        prof = L.profile_read()
    finally:
        L.profile_enable(False)
    return {k: v[0] for k, v in prof.items()}


def _knife_edge_units(params, cfg, x):
    """Hidden units of the first encoder layer whose pre-activation is within float32 rounding
    of zero for some row of this batch: whether such a (row, unit) passes the ReLU is decided by
    the summation order of K products -- the oracle's and the kernel's both round correctly and
    may differ -- and ONE row's contribution then enters or leaves that unit's weight and bias
    gradient (seen: unit 82 of 'tracts', float64 pre-activation -5.7e-10, kernel +9.7e-8).
    {gradient key: [unit, ...]} of the units to leave out of the comparison."""
    out = {}
    if cfg.enc_layers == 0:
        return out
    for name in cfg.names:
        if name not in x:
            continue
        w = "encoders.%s.shared_encoder.0.weight" % name
        b = w.replace("weight", "bias")
        pre = x[name].double() @ params[w].double().t() + params[b].double()
        units = (pre.abs().min(0).values < 2e-6).nonzero().flatten().tolist()
        if units:
            out[w] = out[b] = units
    return out


ROWGROUP_TOPOS = [dict(enc_layers=0), dict(sample_scale=True), dict(enc_layers=0, sample_scale=True)]


@pytest.mark.parametrize("topo", ROWGROUP_TOPOS, ids=lambda t: "-".join("%s%s" % (k[:3], v) for k, v in t.items()))
@pytest.mark.parametrize("method,base,present", [("joint_elbo", C1, None), ("poe", C1, None), ("moe", C1, None),
                                                 ("joint_elbo", C1, ["rois"]), ("joint_elbo", C5, None)])
def test_topologies_that_run_in_the_row_group_kernel(topo, method, base, present, monkeypatch):
    """Two topologies off the default do not take the general chain of launches (csrc:
    rowgroup_route): `num_hidden_layer_encoder = 0` (the default minus its encoder layer: the
    row-group kernel reads x with K = d_m, two launches) and `learn_output_sample_scale` (the
    decoder's logvar head inside the fused launch: a second set of decoder accumulators, a second
    dL/dz contribution, a fourth weight-gradient job), alone or together.  Five free-running
    steps with injected eps against the oracle step by step, the launches counted, and -- same
    inputs, same eps -- against the chain (MOPOE_TOPOLOGY_CHAIN=1)."""
    cfg = mo.Config(method=method, **base, **topo)
    n = 200
    monkeypatch.delenv("MOPOE_TOPOLOGY_CHAIN", raising=False)
    spec, eng = make_engine(cfg)
    _, chain = make_engine(cfg)
    assert spec.general
    rep = Report("row-group topology %s %s %s" % (topo, method, present))
    steps = 5
    xs, tapes = [], []
    for step in range(steps):
        x = mo.make_inputs(cfg.names, cfg.input_dim, n, seed=40 + step, present=present)
        noise = mo.Noise(generator=mo.noise_rng(50 + step))
        params_now = OrderedDict((k, v.cpu().clone()) for k, v in eng.named_params().items())
        st = {"step": OrderedDict((k, step if present is None or k.split(".")[1] in present else 0)
                                  for k in params_now),
              "exp_avg": OrderedDict((k, v.cpu().clone()) for k, v in spec.param_views(eng.exp_avg).items()),
              "exp_avg_sq": OrderedDict((k, v.cpu().clone()) for k, v in spec.param_views(eng.exp_avg_sq).items())}
        edge = _knife_edge_units(params_now, cfg, x)
        out, grads = mo.train_step(params_now, cfg, x, noise, st)
        plan, ws = eng.train_step(x, eps=noise.tape)
        torch.cuda.synchronize()
        p = "step%d/" % step
        compare_forward(rep, spec, eng, plan, ws, out, prefix=p, check_scale=False)
        for k, g in grads.items():
            got, avg = eng.grad_views[k].clone(), spec.param_views(eng.exp_avg)[k].clone()
            for u in edge.get(k, ()):       # (the ReLU decision there is rounding's: either answer is right)
                got[u], avg[u] = g[u], st["exp_avg"][k][u]
            rep.close_scaled(p + "grad/" + k, got, g, TOL["grad"])
            rep.close_scaled(p + "exp_avg/" + k, avg, st["exp_avg"][k], TOL["grad"])
        if cfg.sample_scale:    # the head's (N, d) scale of THIS forward (before the update)
            res = eng.results(plan, ws)
            for k in res["rec"]:
                rep.close(p + "rec/%s/scale" % k, res["rec"][k].scale, out["results"]["rec"][k][1], 2e-5, 1e-6)
        xs.append(x)
        tapes.append(noise.tape)
        eng.check_valid(sync=True)
    rep.finish()
    assert eng.step_count() == steps
    counts = _kernel_counts(make_engine(cfg)[1], xs)
    # (the head's gradient tiles are per decoder job: four modalities, or method poe's two jobs
    #  per modality, do not fit sixteen rows into the LDS with them -- those steps keep the chain)
    fits = not cfg.sample_scale or (base is C1 and method != "poe")
    if not fits:
        assert counts["k_linear"] > 0               # the chain, as before
    elif cfg.enc_layers == 0:      # the row groups alone + the weight gradients
        assert counts["k_latent"] == steps and counts["k_wgrad"] == steps and counts["k_linear"] == 0
        assert counts["k_fused"] == 0
    else:                          # the default's fused launch + the weight gradients
        assert counts["k_fused"] == steps and counts["k_wgrad"] == steps
        assert counts["k_linear"] == 0 and counts["k_latent"] == 0
    # ... against the chain on the same inputs: different summation orders, float32 rounding
    monkeypatch.setenv("MOPOE_TOPOLOGY_CHAIN", "1")
    for x, tape in zip(xs, tapes):
        chain.train_step(x, eps=tape)
    torch.cuda.synchronize()
    chain.check_valid(sync=True)
    assert _kernel_counts(make_engine(cfg)[1], xs[:1])["k_linear"] > 0      # (the knob: the chain ran)
    # (five free-running Adam steps of lr 2e-3 move a parameter by up to 1e-2; an element whose
    #  gradient is ~0 takes its step's SIGN from rounding noise, so two summation orders may part
    #  by a fraction of a step there -- the step-by-step comparison above is the tight one)
    d = (eng.params - chain.params).abs().max().item()
    assert d < 5e-4, d
    assert eng.adam_steps() == chain.adam_steps()

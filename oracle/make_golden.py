"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REFERENCE.

Run in the development container (needs /root/reference):

    python oracle/make_golden.py

For every case below it builds the reference model (through the stub loader
in oracle/ref_harness.py), loads the oracle's deterministic initial weights
into it, replays a recorded noise tape through
`run_epochs.basic_routine_epoch` + `backward` + `torch.optim.Adam.step`, and
stores inputs, noise, every API-visible intermediate, the loss terms, and
digests of gradients / parameters / Adam state.  The fixtures are data only
(inputs and expected outputs); no reference source travels with them.
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import mopoe_oracle as mo  # noqa: E402
import ref_harness as rh  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

C1 = dict(names=["clinical", "rois"], input_dim=[7, 444], style_dim=[3, 20])
C5 = dict(names=["clinical", "rois", "snps", "tracts"],
          input_dim=[7, 444, 128, 64], style_dim=[3, 20])  # -> [3,3,3,3]

CASES = [
    # full-intermediate fixtures (small N)
    dict(case="c1_joint_fact_n32", **C1, method="joint_elbo", factorized=True,
         N=32, steps=3, full=True),
    dict(case="c1_joint_nofact_n37", **C1, method="joint_elbo",
         factorized=False, N=37, steps=2, full=True),
    dict(case="c3_poe_fact_n32", **C1, method="poe", factorized=True, N=32,
         steps=2, full=True),
    dict(case="c3_poe_nofact_n19", **C1, method="poe", factorized=False, N=19,
         steps=1, full=True),
    dict(case="moe_fact_n32", **C1, method="moe", factorized=True, N=32,
         steps=2, full=True),
    dict(case="c5_4mod_joint_fact_n32", **C5, method="joint_elbo",
         factorized=True, N=32, steps=2, full=True),
    dict(case="c5_4mod_moe_nofact_n23", **C5, method="moe", factorized=False,
         N=23, steps=1, full=True),
    dict(case="order_zeta_alpha_n16", names=["zeta", "alpha"],
         input_dim=[7, 12], style_dim=[2, 5], method="joint_elbo",
         factorized=True, N=16, steps=1, full=True),
    # missing-modality batches (MissingModalitySampler yields these)
    dict(case="c1_joint_only_clinical_n24", **C1, method="joint_elbo",
         factorized=True, N=24, steps=2, full=True, present=["clinical"]),
    dict(case="c1_joint_only_rois_n24", **C1, method="joint_elbo",
         factorized=True, N=24, steps=1, full=True, present=["rois"]),
    dict(case="c5_4mod_joint_missing2_n20", **C5, method="joint_elbo",
         factorized=True, N=20, steps=1, full=True,
         present=["clinical", "snps"]),
    dict(case="c3_poe_only_rois_n16", **C1, method="poe", factorized=True,
         N=16, steps=1, full=True, present=["rois"]),
    # BASELINE.json sizes: scalars + digests only
    dict(case="c1_joint_fact_n256", **C1, method="joint_elbo",
         factorized=True, N=256, steps=5, full=False),
    dict(case="c3_poe_fact_n1024", **C1, method="poe", factorized=True,
         N=1024, steps=2, full=False),
    dict(case="c5_4mod_joint_fact_n512", **C5, method="joint_elbo",
         factorized=True, N=512, steps=2, full=False),
]

# Cases added after the first fixture set (their files are new; the ones above are
# regenerated bit-identically).  `present_steps`: the batch of step k holds only these
# modalities -- what MissingModalitySampler feeds the loop (dataset.py:275-354).
# torch.optim.Adam keeps state['step'] per parameter and skips parameters whose grad
# is None, so an encoder / decoder that sat out k steps is k Adam steps behind.
CASES += [
    dict(case="c1_joint_mixed_masks_n24", **C1, method="joint_elbo",
         factorized=True, N=24, steps=6, full=False, every_step=True,
         present_steps=[["clinical", "rois"], ["clinical"], ["clinical", "rois"],
                        ["rois"], ["rois"], ["clinical", "rois"]]),
    dict(case="c5_4mod_mixed_masks_n20", **C5, method="joint_elbo",
         factorized=True, N=20, steps=4, full=False, every_step=True,
         present_steps=[["clinical", "rois", "snps", "tracts"], ["clinical", "snps"],
                        ["rois"], ["clinical", "rois", "snps", "tracts"]]),
    # run_epochs.py:115: method poe without the unimodal ELBOs
    dict(case="c3_poe_nounimodal_n16", **C1, method="poe", factorized=True, N=16,
         steps=2, full=True, poe_unimodal_elbos=False),
    # modalities/modality.py:18-30: --likelihood laplace
    dict(case="c1_joint_laplace_n32", **C1, method="joint_elbo", factorized=True, N=32,
         steps=3, full=True, likelihood="laplace"),
    dict(case="c3_poe_laplace_n19", **C1, method="poe", factorized=True, N=19,
         steps=2, full=True, likelihood="laplace"),
    dict(case="c5_4mod_laplace_n23", **C5, method="joint_elbo", factorized=True, N=23,
         steps=1, full=True, likelihood="laplace"),
]

# The topologies workflow.train_exp exposes beyond its defaults (networks.py:16-20,51-59,
# 66-77): more hidden encoder layers (or none), hidden decoder layers, Dropout behind every
# hidden layer (keep masks recorded like eps), the per-subject output scale head.
CASES += [
    dict(case="t_enc2_dec1_n32", **C1, method="joint_elbo", factorized=True, N=32, steps=3,
         full=True, enc_layers=2, dec_layers=1),
    dict(case="t_enc2_dec1_drop_n32", **C1, method="joint_elbo", factorized=True, N=32,
         steps=3, full=True, enc_layers=2, dec_layers=1, dropout=0.2),
    dict(case="t_sample_scale_n32", **C1, method="joint_elbo", factorized=True, N=32,
         steps=2, full=True, sample_scale=True),
    dict(case="t_sample_scale_dec2_drop_n24", **C1, method="joint_elbo", factorized=False,
         N=24, steps=2, full=True, dec_layers=2, dropout=0.1, sample_scale=True),
    dict(case="t_enc0_n24", **C1, method="joint_elbo", factorized=True, N=24, steps=2,
         full=True, enc_layers=0),
    dict(case="t_poe_enc2_dec1_drop_n19", **C1, method="poe", factorized=True, N=19, steps=2,
         full=True, enc_layers=2, dec_layers=1, dropout=0.25),
    dict(case="t_moe_enc3_n21", **C1, method="moe", factorized=True, N=21, steps=1,
         full=True, enc_layers=3, dec_layers=1, likelihood="laplace"),
    dict(case="t_c5_enc3_dec2_drop_n20", **C5, method="joint_elbo", factorized=True, N=20,
         steps=2, full=True, enc_layers=3, dec_layers=2, dropout=0.15),
    dict(case="t_c5_missing_enc2_dec1_n20", **C5, method="joint_elbo", factorized=True, N=20,
         steps=1, full=True, enc_layers=2, dec_layers=1, dropout=0.2,
         present=["rois", "tracts"]),
    dict(case="t_enc2_dec1_drop_n256", **C1, method="joint_elbo", factorized=True, N=256,
         steps=3, full=False, enc_layers=2, dec_layers=1, dropout=0.2),
]

# forward-only variants (BaseMMVae.forward flags), on the c1 model
FWD_CASES = [
    dict(case="fwd_c1_nosample_n16", **C1, method="joint_elbo",
         factorized=True, N=16, sample_latents=False, use_expert=None),
    dict(case="fwd_c1_expert_n16", **C1, method="joint_elbo", factorized=True,
         N=16, sample_latents=True, use_expert="clinical_rois"),
    dict(case="fwd_c1_expert_nosample_n16", **C1, method="joint_elbo",
         factorized=True, N=16, sample_latents=False, use_expert="rois"),
]


def np32(t):
    return t.detach().to(torch.float32).numpy().copy()


def digest(t):
    """sum, abs-sum, l2 (float64) + strided sample (<= 64 values; the stride
    rule is mo.digest_stride) of a tensor."""
    f = t.detach().double().reshape(-1)
    stats = np.array([f.sum().item(), f.abs().sum().item(),
                      f.pow(2).sum().sqrt().item()], dtype=np.float64)
    flat = t.detach().float().reshape(-1)
    sample = flat[::mo.digest_stride(flat.numel())]
    return stats, sample.numpy().copy()


def put_digest(store, prefix, named):
    for k, v in named.items():
        s, sample = digest(v)
        store[prefix + "/" + k + "/stats"] = s
        store[prefix + "/" + k + "/sample"] = sample


def make_inputs(c, seed):
    return mo.make_inputs(c["names"], c["input_dim"], c["N"], seed,
                          present=c.get("present"))


def checksum(t):
    f = t.detach().double().reshape(-1)
    return np.array([f.sum().item(), f.abs().sum().item()], dtype=np.float64)


def build(ns, c, seed=0):
    topo = {k: c[k] for k in ("enc_layers", "dec_layers", "dropout", "sample_scale") if k in c}
    flags = rh.make_flags(c["input_dim"],
                          mo.Config(c["names"], c["input_dim"], c["style_dim"],
                                    factorized=c["factorized"]).style_dim,
                          method=c["method"], factorized=c["factorized"],
                          poe_unimodal_elbos=c.get("poe_unimodal_elbos", True),
                          likelihood=c.get("likelihood", "normal"), **topo)
    exp = rh.build_experiment(ns, flags, c["names"])
    cfg = mo.Config(c["names"], c["input_dim"], c["style_dim"],
                    method=c["method"], factorized=c["factorized"],
                    poe_unimodal_elbos=c.get("poe_unimodal_elbos", True),
                    likelihood=c.get("likelihood", "normal"), **topo)
    init = mo.init_params(cfg, seed)
    missing, unexpected = exp.models.load_state_dict(init, strict=True)
    assert not missing and not unexpected
    return exp, cfg


def flatten_results(store, prefix, out, full, method):
    res = out["results"] if "results" in out else out
    lat = res["latents"]
    if "total_loss" in out:
        store[prefix + "/total_loss"] = np32(out["total_loss"])
        for k, v in out["log_probs"].items():
            store[prefix + "/log_probs/" + k] = np32(v)
        for k, v in out["klds"].items():
            store[prefix + "/klds/" + k] = np32(v)
    store[prefix + "/joint_divergence"] = np32(res["joint_divergence"])
    store[prefix + "/individual_divs"] = np32(res["individual_divs"])
    store[prefix + "/weights"] = np32(lat["weights"])
    if not full:
        return
    for k, (mu, lv) in lat["modalities"].items():
        if mu is not None:
            store[prefix + "/modalities/" + k + "/mu"] = np32(mu)
            store[prefix + "/modalities/" + k + "/logvar"] = np32(lv)
    for k, (mu, lv) in lat["subsets"].items():
        store[prefix + "/subsets/" + k + "/mu"] = np32(mu)
        store[prefix + "/subsets/" + k + "/logvar"] = np32(lv)
    if method == "joint_elbo":
        # mus/logvars are the stacked subset distributions (checked here,
        # not stored twice)
        assert torch.equal(lat["mus"], torch.stack(
            [v[0] for v in lat["subsets"].values()]))
        assert torch.equal(lat["logvars"], torch.stack(
            [v[1] for v in lat["subsets"].values()]))
    else:
        store[prefix + "/mus"] = np32(lat["mus"])
        store[prefix + "/logvars"] = np32(lat["logvars"])
    store[prefix + "/joint/mu"] = np32(lat["joint"][0])
    store[prefix + "/joint/logvar"] = np32(lat["joint"][1])
    for k, dist in res["rec"].items():
        store[prefix + "/rec/" + k + "/loc"] = np32(dist.loc)
        store[prefix + "/rec/" + k + "/scale"] = np32(dist.scale)


def run_case(ns, c):
    exp, cfg = build(ns, c)
    model = exp.models
    model.train()
    opt = torch.optim.Adam(list(model.parameters()), lr=cfg.lr,
                           betas=cfg.betas)
    x = make_inputs(c, seed=1234)
    store = OrderedDict()
    store["meta"] = np.array(json.dumps(
        {k: v for k, v in c.items()}, sort_keys=True))
    store["subset_keys"] = np.array(json.dumps(list(exp.subsets.keys())))
    for k, v in x.items():
        store["in/checksum/" + k] = checksum(v)
        if c["full"]:
            store["in/x/" + k] = np32(v)
    for step in range(c["steps"]):
        tape = rh.NoiseTape(model, generator=mo.noise_rng(4321 + step))
        masks = rh.MaskTape(model, generator=mo.noise_rng(8765 + step))
        present = c["present_steps"][step] if "present_steps" in c else list(x)
        batch = (OrderedDict((k, v.double()) for k, v in x.items() if k in present),
                 None, {})
        out = ns.run_epochs.basic_routine_epoch(exp, 0, batch)
        for i, e in enumerate(tape.tape):
            store["noise_checksum/%d/%d" % (step, i)] = checksum(e)
            if c["full"]:
                store["noise/%d/%d" % (step, i)] = np32(e)
        for i, e in enumerate(masks.tape):      # dropout keep masks, bit-packed
            store["mask_checksum/%d/%d" % (step, i)] = checksum(e)
            if c["full"]:
                store["mask/%d/%d" % (step, i)] = np.packbits(e.numpy().astype(np.uint8))
                store["mask_shape/%d/%d" % (step, i)] = np.array(e.shape)
        opt.zero_grad()
        out["total_loss"].backward()
        if step == 0:
            flatten_results(store, "step0", out, c["full"], c["method"])
            put_digest(store, "step0/grads", OrderedDict(
                (k, p.grad) for k, p in model.named_parameters()
                if p.grad is not None))
            store["step0/grad_none"] = np.array(json.dumps(
                [k for k, p in model.named_parameters() if p.grad is None]))
        else:
            store["step%d/total_loss" % step] = np32(out["total_loss"])
        if c.get("every_step"):
            store["step%d/grad_none" % step] = np.array(json.dumps(
                [k for k, p in model.named_parameters() if p.grad is None]))
        opt.step()
        if step in (0, c["steps"] - 1) or c.get("every_step"):
            put_digest(store, "after%d/params" % (step + 1),
                       OrderedDict(model.named_parameters()))
    st = opt.state_dict()["state"]
    names = [k for k, _ in model.named_parameters()]
    put_digest(store, "final/exp_avg", OrderedDict(
        (names[i], s["exp_avg"]) for i, s in st.items()))
    put_digest(store, "final/exp_avg_sq", OrderedDict(
        (names[i], s["exp_avg_sq"]) for i, s in st.items()))
    # torch's per-parameter step counts (a parameter never stepped has no state)
    if c.get("every_step"):
        store["final/adam_steps"] = np.array(json.dumps(
            {names[i]: int(s["step"]) for i, s in st.items()}))
    return store


def run_fwd_case(ns, c):
    exp, cfg = build(ns, c)
    model = exp.models
    model.eval()
    x = make_inputs(c, seed=99)
    store = OrderedDict()
    store["meta"] = np.array(json.dumps(c, sort_keys=True))
    for k, v in x.items():
        store["in/x/" + k] = np32(v)
    tape = rh.NoiseTape(model, generator=mo.noise_rng(7))
    with torch.no_grad():
        res = model(OrderedDict(x), sample_latents=c["sample_latents"],
                    use_expert=c["use_expert"])
    for i, e in enumerate(tape.tape):
        store["noise/0/%d" % i] = np32(e)
    flatten_results(store, "step0", res, True, c["method"])
    return store


def l0_vectors(ns):
    """Known-answer vectors for the free functions of section 8b."""
    g = torch.Generator().manual_seed(5)
    store = OrderedDict()
    for E in (1, 2, 3, 5):
        mu = torch.randn(E, 9, 20, generator=g)
        lv = torch.randn(E, 9, 20, generator=g) * 1.5
        pm, plv = ns.mm_div.poe(mu, lv)
        store["poe/%d/mu" % E] = np32(mu)
        store["poe/%d/logvar" % E] = np32(lv)
        store["poe/%d/out_mu" % E] = np32(pm)
        store["poe/%d/out_logvar" % E] = np32(plv)
    mu = torch.randn(33, 20, generator=g)
    lv = torch.randn(33, 20, generator=g)
    store["kl/mu"] = np32(mu)
    store["kl/logvar"] = np32(lv)
    store["kl/out"] = np32(ns.kl_div.calc_kl_divergence(mu, lv))
    store["kl/out_norm"] = np32(
        ns.kl_div.calc_kl_divergence(mu, lv, norm_value=33))
    flags = rh.make_flags([7, 444], [3, 20])
    for K, N in ((3, 256), (15, 512), (3, 37), (7, 5), (2, 1), (1, 8)):
        mus = torch.randn(K, N, 4, generator=g)
        lvs = torch.randn(K, N, 4, generator=g)
        w = ns.utils.reweight_weights((1 / float(K)) * torch.ones(K))
        m_sel, l_sel = ns.utils.mixture_component_selection(flags, mus, lvs, w)
        store["mix/%d_%d/mus" % (K, N)] = np32(mus)
        store["mix/%d_%d/logvars" % (K, N)] = np32(lvs)
        store["mix/%d_%d/out_mu" % (K, N)] = np32(m_sel)
        store["mix/%d_%d/out_logvar" % (K, N)] = np32(l_sel)
        gd, klds = ns.mm_div.calc_group_divergence_moe(flags, mus, lvs, w,
                                                       normalization=N)
        store["mix/%d_%d/group_div" % (K, N)] = np32(gd)
        store["mix/%d_%d/klds" % (K, N)] = np32(klds)
    return store


def sampler_vectors(ns):
    """MissingModalitySampler of the reference on a synthetic cohort: which
    subjects land in which batch, for a fixed numpy seed."""
    import importlib
    import types
    from itertools import chain, combinations
    ref_ds = importlib.import_module("multimodal_cohort.dataset")
    rng = np.random.RandomState(3)
    n = 101
    has = {"clinical": rng.rand(n) > 0.15, "rois": rng.rand(n) > 0.25}
    has["clinical"] |= ~has["rois"]          # every subject has at least one block
    mods = ["clinical", "rois"]
    subsets = list(chain.from_iterable(combinations(mods, k) for k in range(1, 3)))
    per_subset = [[] for _ in subsets]
    for i in range(n):
        present = tuple(m for m in mods if has[m][i])
        per_subset[subsets.index(present)].append(i)
    ds = types.SimpleNamespace(modality_subsets=subsets, idx_per_modality_subset=per_subset,
                               metadata=None)
    store = OrderedDict()
    store["has/clinical"] = has["clinical"]
    store["has/rois"] = has["rois"]
    def make_sampler(bs):
        # the reference ctor calls Sampler.__init__(dataset), which torch 2.10
        # no longer accepts (it is pinned to torch 1.13): set its fields directly
        smp = ref_ds.MissingModalitySampler.__new__(ref_ds.MissingModalitySampler)
        smp.dataset, smp.indices, smp.batch_size = ds, None, bs
        smp.stratify, smp.discretize, smp.seed = None, None, 42
        return smp

    for seed, bs in ((7, 16), (11, 32), (5, 200)):
        np.random.seed(seed)
        batches = list(make_sampler(bs))
        store["batches/%d_%d" % (seed, bs)] = np.array(json.dumps(
            [[int(i) for i in b] for b in batches]))
        store["len/%d_%d" % (seed, bs)] = np.array(len(make_sampler(bs)))
    return store


def scaler_vectors(ns):
    """The input scaling of the reference on a synthetic cohort: the StandardScaler its
    MultimodalExperiment.set_scalers fits (experiment.py:146-166) and what its
    per-sample on-the-fly transform chain delivers (experiment.py:228-232:
    unsqueeze(0) -> scaler.transform -> ToTensor -> squeeze), cast as the training
    loop casts it (run_epochs.py:86, .float())."""
    import importlib
    import types
    ref_exp = importlib.import_module("multimodal_cohort.experiment")
    tv = sys.modules["torchvision.transforms"]

    class Compose:      # the two torchvision transforms the chain uses, on tensors/arrays
        def __init__(self, fns):
            self.fns = fns

        def __call__(self, v):
            for f in self.fns:
                v = f(v)
            return v

    class ToTensor:
        def __call__(self, a):       # (H, W) ndarray -> (1, H, W) tensor
            return torch.from_numpy(np.asarray(a))[None]

    tv.Compose, tv.ToTensor = Compose, ToTensor
    rng = np.random.RandomState(11)
    n = 60
    has = {"clinical": rng.rand(n) > 0.2, "rois": rng.rand(n) > 0.25}
    has["clinical"] |= ~has["rois"]
    dims = {"clinical": 7, "rois": 12}
    data, idx = {}, {}
    for mod in ("clinical", "rois"):
        rows = np.flatnonzero(has[mod])
        data[mod] = rng.randn(len(rows), dims[mod]) * rng.uniform(0.5, 40.0, dims[mod]) + \
            rng.uniform(-30.0, 30.0, dims[mod])
        perm = rng.permutation(len(rows))
        col = np.empty(n, dtype=object)
        col[:] = None
        for k, subj in enumerate(rows):
            col[subj] = int(perm[k])
        idx[mod] = col
    data["clinical"][:, 3] = 2.5          # a constant feature: scale stays 1
    samples = []                          # what the reference's dataset[i] returns
    for i in range(n):
        samples.append(({mod: torch.tensor(data[mod][idx[mod][i]]) for mod in dims
                         if idx[mod][i] is not None}, 0, {}))
    fake = types.SimpleNamespace(mod_names=["clinical", "rois"])
    scalers = ref_exp.MultimodalExperiment.set_scalers(fake, samples)
    store = OrderedDict()
    for mod in dims:
        store["data/" + mod] = data[mod]
        store["idx/" + mod] = np.array([-1 if r is None else r for r in idx[mod]], dtype=np.int64)
        store["mean/" + mod] = scalers[mod].mean_
        store["scale/" + mod] = scalers[mod].scale_
        chain = tv.Compose([lambda x: x.unsqueeze(0), scalers[mod].transform, tv.ToTensor(),
                            torch.squeeze])
        rows = [chain(s[0][mod]).float().numpy() for s in samples if mod in s[0]]
        store["transformed/" + mod] = np.stack(rows)      # subjects that have it, in order
    return store


def main():
    ns = rh.import_reference()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    total = 0
    only = sys.argv[1:]            # optional: case names to (re)generate
    if only:
        if "scaler" in only:
            np.savez_compressed(os.path.join(GOLDEN_DIR, "scaler.npz"), **scaler_vectors(ns))
            print("scaler.npz written")
        for c in CASES:
            if c["case"] in only:
                store = run_case(ns, c)
                path = os.path.join(GOLDEN_DIR, c["case"] + ".npz")
                np.savez_compressed(path, **store)
                print("%-32s %8d B  loss=%s" % (c["case"], os.path.getsize(path),
                                                store["step0/total_loss"]))
        return
    for c in CASES:
        store = run_case(ns, c)
        path = os.path.join(GOLDEN_DIR, c["case"] + ".npz")
        np.savez_compressed(path, **store)
        total += os.path.getsize(path)
        print("%-32s %8d B  loss=%s" % (c["case"], os.path.getsize(path),
                                        store["step0/total_loss"]))
    for c in FWD_CASES:
        store = run_fwd_case(ns, c)
        path = os.path.join(GOLDEN_DIR, c["case"] + ".npz")
        np.savez_compressed(path, **store)
        total += os.path.getsize(path)
        print("%-32s %8d B" % (c["case"], os.path.getsize(path)))
    store = l0_vectors(ns)
    path = os.path.join(GOLDEN_DIR, "l0_functions.npz")
    np.savez_compressed(path, **store)
    total += os.path.getsize(path)
    store = sampler_vectors(ns)
    path = os.path.join(GOLDEN_DIR, "sampler.npz")
    np.savez_compressed(path, **store)
    total += os.path.getsize(path)
    store = scaler_vectors(ns)
    path = os.path.join(GOLDEN_DIR, "scaler.npz")
    np.savez_compressed(path, **store)
    total += os.path.getsize(path)
    print("total %d B" % total)


if __name__ == "__main__":
    main()

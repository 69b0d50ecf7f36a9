"""Helpers shared by the oracle and HIP parity tests: load a fixture written
by oracle/make_golden.py and rebuild its inputs (the oracle is the checker;
see oracle/mopoe_oracle.py's header)."""
import glob
import json
import os
from collections import OrderedDict

import numpy as np
import torch

import mopoe_oracle as mo

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names(prefix=None, fwd=False):
    names = sorted(os.path.basename(p)[:-4]
                   for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    names = [n for n in names if n not in ("l0_functions", "sampler", "scaler")]
    names = [n for n in names if n.startswith("fwd_") == fwd]
    if prefix:
        names = [n for n in names if n.startswith(prefix)]
    return names


class Fixture:
    def __init__(self, case):
        self.case = case
        self.z = np.load(os.path.join(GOLDEN_DIR, case + ".npz"),
                         allow_pickle=False)
        self.meta = json.loads(str(self.z["meta"]))
        m = self.meta
        self.cfg = mo.Config(m["names"], m["input_dim"], m["style_dim"],
                             method=m["method"], factorized=m["factorized"],
                             poe_unimodal_elbos=m.get("poe_unimodal_elbos", True),
                             likelihood=m.get("likelihood", "normal"),
                             **{k: m[k] for k in ("enc_layers", "dec_layers", "dropout",
                                                  "sample_scale") if k in m})
        self.N = m["N"]
        self.steps = m.get("steps", 1)
        self.full = m.get("full", True)
        self.present = m.get("present") or m["names"]
        # batches whose modality set changes from step to step (mixed-mask cases)
        self.present_steps = m.get("present_steps")
        self.every_step = bool(m.get("every_step"))

    def keys(self, prefix):
        return [k for k in self.z.files if k.startswith(prefix)]

    def get(self, key):
        return torch.from_numpy(np.asarray(self.z[key]))

    def has(self, key):
        return key in self.z.files

    def inputs(self):
        seed = 99 if self.case.startswith("fwd_") else 1234
        x = mo.make_inputs(self.meta["names"], self.meta["input_dim"], self.N,
                           seed, present=self.meta.get("present"))
        for k, v in x.items():
            if self.has("in/x/" + k):
                assert torch.equal(v, self.get("in/x/" + k)), \
                    "numpy PCG64 stream drifted from the fixture"
            else:
                cs = self.z["in/checksum/" + k]
                f = v.double().reshape(-1)
                assert abs(f.sum().item() - cs[0]) < 1e-6 * max(1, abs(cs[0]))
                assert abs(f.abs().sum().item() - cs[1]) < 1e-6 * cs[1]
        return x

    def inputs_at(self, step):
        """The batch of step `step`: the case's inputs restricted to the modalities
        that step's batch holds."""
        x = self.inputs()
        if self.present_steps is None:
            return x
        return OrderedDict((k, v) for k, v in x.items()
                           if k in self.present_steps[step])

    def noise_seed(self, step):
        return (7 if self.case.startswith("fwd_") else 4321 + step)

    def noise(self, step):
        """mo.Noise that regenerates the fixture's eps stream (and the stream of its
        dropout keep masks) for `step`."""
        return mo.Noise(generator=mo.noise_rng(self.noise_seed(step)),
                        mask_generator=mo.noise_rng(8765 + step))

    def check_noise(self, step, noise):
        for i, e in enumerate(noise.tape):
            if self.has("noise/%d/%d" % (step, i)):
                assert torch.equal(e, self.get("noise/%d/%d" % (step, i)))
            elif self.has("noise_checksum/%d/%d" % (step, i)):
                cs = self.z["noise_checksum/%d/%d" % (step, i)]
                f = e.double().reshape(-1)
                assert abs(f.sum().item() - cs[0]) < 1e-6 * max(1, abs(cs[0]))
        for i, e in enumerate(noise.mask_tape):      # dropout keep masks (bit-packed)
            if self.has("mask/%d/%d" % (step, i)):
                shape = tuple(int(v) for v in self.z["mask_shape/%d/%d" % (step, i)])
                bits = np.unpackbits(self.z["mask/%d/%d" % (step, i)])[:e.numel()]
                assert torch.equal(e, torch.from_numpy(bits.astype(np.float32)).reshape(shape))
            else:
                assert self.has("mask_checksum/%d/%d" % (step, i)), "more masks than recorded"
                assert e.double().sum().item() == self.z["mask_checksum/%d/%d" % (step, i)][0]
        nmask = len([k for k in self.z.files if k.startswith("mask_checksum/%d/" % step)])
        assert nmask == len(noise.mask_tape), "dropout masks: %d drawn, %d recorded" % (
            len(noise.mask_tape), nmask)


def assert_close(a, b, rtol, atol, what=""):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    if not bool((err <= tol).all()):
        i = int((err - tol).argmax())
        raise AssertionError(
            "%s: max abs err %.3e (at %d: got %.8g want %.8g), rtol %g atol %g"
            % (what, err.max().item(), i, a.reshape(-1)[i].item(),
               b.reshape(-1)[i].item(), rtol, atol))


def digest_of(t):
    f = t.detach().cpu().double().reshape(-1)
    stats = torch.tensor([f.sum().item(), f.abs().sum().item(),
                          f.pow(2).sum().sqrt().item()], dtype=torch.float64)
    flat = t.detach().cpu().float().reshape(-1)
    return stats, flat[::mo.digest_stride(flat.numel())]


def check_digests(fx, prefix, named, rtol, atol, skip_missing=False):
    """Compare {name: tensor} against the fixture's digests under prefix."""
    want = sorted(set(k[len(prefix) + 1:].rsplit("/", 1)[0]
                      for k in fx.keys(prefix + "/")))
    if not skip_missing:
        assert sorted(named.keys()) == want, (sorted(named.keys()), want)
    for name in want:
        if name not in named:
            continue
        stats, sample = digest_of(named[name])
        w_stats = fx.get(prefix + "/" + name + "/stats")
        w_sample = fx.get(prefix + "/" + name + "/sample")
        assert_close(sample, w_sample, rtol, atol, prefix + "/" + name)
        # l2 norm is a well-conditioned digest; sum is not (cancellation)
        assert_close(stats[2], w_stats[2], 10 * rtol,
                     atol * max(1.0, sample.numel() ** 0.5),
                     prefix + "/" + name + "/l2")
        assert_close(stats[1], w_stats[1], 10 * rtol,
                     atol * max(1, named[name].numel()),
                     prefix + "/" + name + "/abssum")

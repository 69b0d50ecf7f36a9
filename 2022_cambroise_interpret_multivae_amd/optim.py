"""torch.optim.Optimizer front of mopoe_adam_step: `optimizer.zero_grad();
total_loss.backward(); optimizer.step()` keeps its reference shape
(run_epochs.py:180-182, experiment.py:256-279) while the update runs as one
HIP kernel over the flat parameter buffer."""
import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=0.002, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self._sync()
        self.world = 1     # > 1: engine.grads holds the sum over that many ranks

    def _sync(self):
        """The kernels' Adam scalars follow param_groups (a scheduler may change lr)."""
        g = self.param_groups[0]
        now = (g["lr"], g["betas"][0], g["betas"][1], g["eps"])
        if getattr(self, "_synced", None) != now:
            self.model.engine.adam = L.Adam(*now)
            self._synced = now

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        self._sync()
        eng = self.model.engine
        eng.adam_step(present_mask=eng.last_present_mask, world=self.world)

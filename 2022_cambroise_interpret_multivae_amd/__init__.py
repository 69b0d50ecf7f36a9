"""MI355X-native MoPoE-VAE training hot path (drop-in for the reference's
run_epochs.py step).  The package name starts with a digit, so import it with
importlib.import_module("2022_cambroise_interpret_multivae_amd") or through
the `mopoe_amd` alias module at the repository root."""
from . import _lib  # noqa: F401  (fails loudly if libmopoe_hip.so is missing)
from .plan import ModelSpec, StepPlan  # noqa: F401
from .engine import MoPoEEngine  # noqa: F401
from . import comm  # noqa: F401

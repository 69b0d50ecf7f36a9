"""Mirror of the reference's divergence_measures/kl_div.py (hot-path part)."""
from .. import ops


def calc_kl_divergence(mu0, logvar0, mu1=None, logvar1=None, norm_value=None):
    """KL(N(mu0, e^logvar0) || N(0, I)) summed over all elements, divided by
    norm_value if given (reference divergence_measures/kl_div.py:7-14).  The
    two-Gaussian form (mu1, logvar1) is not on the hot path."""
    if mu1 is not None or logvar1 is not None:
        raise NotImplementedError("calc_kl_divergence with (mu1, logvar1) is only "
                                  "used by method jsd, which is outside the hot path")
    return ops.kl_divergence(mu0, logvar0, norm_value)

"""Mirror of the reference's multimodal_cohort/networks/VAE.py."""
from ...utils.BaseMMVae import BaseMMVae


class VAE(BaseMMVae):
    def __init__(self, flags, modalities, subsets):
        super().__init__(flags, modalities, subsets)

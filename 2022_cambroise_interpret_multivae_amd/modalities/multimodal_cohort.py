"""Mirror of the reference's modalities/multimodal_cohort.py."""
import torch

from .modality import Modality


class Clinical(Modality):
    def __init__(self, n_scores, enc, dec, class_dim, style_dim, lhood_name):
        super().__init__("clinical", enc, dec, class_dim, style_dim, lhood_name)
        self.data_size = torch.Size([n_scores])
        self.gen_quality_eval = True
        self.file_suffix = ".npy"
        self.names_file = "clinical_names.npy"


class Rois(Modality):
    def __init__(self, n_rois, enc, dec, class_dim, style_dim, lhood_name):
        super().__init__("rois", enc, dec, class_dim, style_dim, lhood_name)
        self.data_size = torch.Size([n_rois])
        self.gen_quality_eval = True
        self.file_suffix = ".npy"
        self.names_file = "rois_names.npy"

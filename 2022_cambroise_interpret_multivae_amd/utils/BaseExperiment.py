"""Mirror of the part of the reference's utils/BaseExperiment.py the hot path
needs: the ordered powerset of modalities."""
from collections import OrderedDict
from itertools import chain, combinations


def set_subsets(modalities):
    """{'': [], 'clinical': [mod], ..., 'clinical_rois': [mod, mod]} -- keys are
    '_'.join(sorted(names)), in powerset order by subset size (reference
    utils/BaseExperiment.py:58-79)."""
    xs = list(modalities)
    subsets = OrderedDict()
    for mod_names in chain.from_iterable(combinations(xs, n) for n in range(len(xs) + 1)):
        subsets["_".join(sorted(mod_names))] = [modalities[n] for n in sorted(mod_names)]
    return subsets


class BaseExperiment:
    def set_subsets(self):
        mods = self.modalities
        if type(mods) is list:
            mods = mods[0]
        return set_subsets(mods)

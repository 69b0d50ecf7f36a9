"""Mirror of the reference's multimodal_cohort/dataset.py for the input side
of the hot path.

* `MultimodalDataset`: the reference's per-subject view over per-modality
  blocks (`idx_per_mod[mod][subject]` = row of that block, or None when the
  subject lacks the modality; reference dataset.py:15-147), built from arrays
  in memory or from the on-disk layout the reference's fetchers write
  (`multiblock_idx_{train,test}.npz`, `{mod}_data.npy`).
* `MissingModalitySampler`: batches whose samples all have the SAME set of
  modalities, complete batches shuffled before incomplete ones, drawn with
  np.random.choice without replacement (reference dataset.py:275-354; the
  stratified variant needs `iterstrat`, which is not in the image).
* `ResidentCohort`: the MI355X-native input path.  The reference pushes every
  sample through `torch.tensor -> StandardScaler.transform -> ...` in freshly
  forked DataLoader workers; here each block is scaled once, uploaded once,
  and a batch is a vector of row indices -- the kernels gather the rows
  themselves (`row_index` of the C ABI), so an epoch moves no sample data.
"""
import copy
from itertools import chain, combinations

import numpy as np
import torch

from .. import _lib as L
from ..engine import IndexBatch


class MultimodalDataset(torch.utils.data.Dataset):
    def __init__(self, data, idx_per_mod, metadata=None, indices=None,
                 on_the_fly_transform=None):
        self.data = {k: np.asarray(v) for k, v in data.items()}
        self.idx_per_mod = {k: np.asarray(v, dtype=object) for k, v in idx_per_mod.items()}
        self.modalities = list(self.idx_per_mod)
        n_samples = [len(self.idx_per_mod[key]) for key in self.modalities]
        if len(set(n_samples)) != 1:
            raise ValueError("All modalities do not have the same number of samples.")
        self.metadata = metadata
        if metadata is not None and len(metadata) != n_samples[0]:
            raise ValueError("The data and metadata do not have the same number of samples.")
        self.n_samples = n_samples[0]
        self.indices = indices
        self.on_the_fly_transform = on_the_fly_transform
        self.modality_subsets = list(chain.from_iterable(
            combinations(self.modalities, n) for n in range(1, len(self.modalities) + 1)))
        self.idx_per_modality_subset = self.compute_idx_per_modality_subset()

    @classmethod
    def from_files(cls, idx_path, metadata=None, indices=None):
        """The on-disk layout of the reference (dataset.py:23,54-90): an .npz of
        object arrays (row index or None per subject) next to `{mod}_data.npy`.
        The index file is the user's own data; it holds Python `None`s, hence
        allow_pickle."""
        idx_per_mod = dict(np.load(idx_path, allow_pickle=True))
        data_path = idx_path.replace("idx", "data").replace(".npz", ".npy")
        data_path = data_path.replace("_train", "").replace("_test", "")
        data = {mod: np.load(data_path.replace("multiblock", mod), mmap_mode="r")
                for mod in idx_per_mod}
        return cls(data, idx_per_mod, metadata=metadata, indices=indices)

    def __len__(self):
        return len(self.indices) if self.indices is not None else self.n_samples

    def _true_idx(self, idx):
        return self.indices[idx] if self.indices is not None else idx

    def __getitem__(self, idx):
        idx = self._true_idx(idx)
        ret = {}
        for mod in self.modalities:
            row = self.idx_per_mod[mod][idx]
            if row is not None:
                v = torch.as_tensor(np.array(self.data[mod][int(row)]))
                t = self.on_the_fly_transform
                if t is not None:
                    v = t[mod](v) if isinstance(t, dict) and mod in t else (
                        v if isinstance(t, dict) else t(v))
                ret[mod] = v
        metadata = {}
        if self.metadata is not None:
            metadata = self.metadata.iloc[idx].to_dict()
        label = metadata["asd"] - 1 if "asd" in metadata else 0
        return ret, label, metadata

    def compute_idx_per_modality_subset(self):
        """reference dataset.py:128-144"""
        out = [[] for _ in self.modality_subsets]
        for idx in range(len(self)):
            true_idx = self._true_idx(idx)
            mods = tuple(m for m in self.modalities
                         if self.idx_per_mod[m][true_idx] is not None)
            for sub_idx, subset in enumerate(self.modality_subsets):
                if set(subset) == set(mods):
                    out[sub_idx].append(idx)
                    break
        return out

    def get_modality_proportions(self):
        return [len(s) / len(self) for s in self.idx_per_modality_subset]


class MissingModalitySampler(torch.utils.data.Sampler):
    """reference dataset.py:275-354 (non-stratified path).

    `world` > 1 (data-parallel replicas, one process per GPU): the epoch is the
    reference's own draw for batches of `batch_size * world` samples -- made by rank 0
    and shared, so every rank deals from the same schedule -- and rank `rank` yields
    its contiguous share of every global batch.  All ranks therefore step on batches
    that hold the SAME modalities, which torch's per-parameter Adam step counts need
    (parallel.py).  `loss_scales[k]` is the weight n_r * world / n_global of this rank's
    k-th batch (1.0 when the global batch splits evenly; a rank left without a sample by
    a ragged batch re-uses one with weight 0)."""

    def __init__(self, dataset, batch_size, indices=None, stratify=None,
                 discretize=None, seed=42, world=1, rank=0):
        if stratify is not None:
            raise NotImplementedError("stratified batches need iterstrat "
                                      "(MultilabelStratifiedKFold), absent from this image")
        if not 0 <= rank < world:
            raise ValueError("rank %d outside world %d" % (rank, world))
        self.dataset = dataset
        self.indices = indices
        self.batch_size = batch_size
        self.seed = seed
        self.world = world
        self.rank = rank
        self.loss_scales = []

    def __len__(self):
        size = self.batch_size * self.world
        return sum((len(idx) + size - 1) // size
                   for idx in self.dataset.idx_per_modality_subset)

    def _subset_arrays(self):
        """The per-subset index lists as (prefix offsets, one int64 array), built once."""
        cache = getattr(self.dataset, "_subset_flat", None)
        if cache is None or cache[0] is not self.dataset.idx_per_modality_subset:
            parts = [np.asarray(p, dtype=np.int64) for p in self.dataset.idx_per_modality_subset]
            begin = np.zeros(len(parts) + 1, dtype=np.int64)
            begin[1:] = np.cumsum([len(p) for p in parts])
            items = np.concatenate(parts) if parts else np.zeros(0, np.int64)
            cache = (self.dataset.idx_per_modality_subset, begin, np.ascontiguousarray(items))
            self.dataset._subset_flat = cache
        return cache[1], cache[2]

    @staticmethod
    def draw_from(state, begin, items, batch_size):
        """One epoch drawn from a GIVEN legacy RandomState state tuple (np.random.get_state()):
        mopoe_sampler_epoch -- MT19937, random_interval and the shuffle of numpy's legacy
        generator restated in C, bit for bit the draws of `draw_numpy`.  Returns (the state
        after the draws, items, batch offsets, modality subset of every batch).  Touches no
        global state, so a helper thread may run it ahead of time."""
        import ctypes as C
        if state[0] != "MT19937":
            raise ValueError("the legacy global RandomState is MT19937")
        key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
        pos = C.c_int32(int(state[2]))
        bs = int(batch_size)
        lens = np.diff(begin)
        max_batches = int(((lens + bs - 1) // bs).sum())
        out_items = np.empty(len(items), dtype=np.int64)
        out_begin = np.zeros(max_batches + 1, dtype=np.int64)
        out_subset = np.zeros(max(max_batches, 1), dtype=np.int32)
        nb = C.c_int64(0)
        L.check(L.lib.mopoe_sampler_epoch(
            key.ctypes.data, C.byref(pos), len(lens), begin.ctypes.data, items.ctypes.data, bs,
            out_items.ctypes.data, out_begin.ctypes.data, out_subset.ctypes.data, C.byref(nb)),
            "mopoe_sampler_epoch")
        n = int(nb.value)
        return (("MT19937", key, int(pos.value)) + tuple(state[3:]), out_items,
                out_begin[:n + 1], out_subset[:n])

    def draw(self, batch_size):
        """The reference's epoch for this batch size, off numpy's global legacy stream (which
        is left where the reference's own draws would have left it)."""
        begin, items = self._subset_arrays()
        state, out_items, out_begin, _ = self.draw_from(np.random.get_state(), begin, items,
                                                        batch_size)
        np.random.set_state(state)
        return [out_items[out_begin[k]:out_begin[k + 1]] for k in range(len(out_begin) - 1)]

    def draw_numpy(self, batch_size):
        """The same epoch through numpy itself (the C draw's checker in the tests)."""
        indices, complete, incomplete = [], [], []
        batch_idx = 0
        for idx, _ in enumerate(self.dataset.modality_subsets):
            # the reference keeps a list and list.remove()s every drawn index (O(n^2):
            # 2.6 s per epoch at 16k samples); an order-preserving mask over an array
            # leaves the same remainder, so np.random.choice draws the same batches
            cache = getattr(self.dataset, "_subset_arrays", None)
            if cache is None or cache[0] is not self.dataset.idx_per_modality_subset:
                cache = (self.dataset.idx_per_modality_subset,
                         [np.asarray(p, dtype=np.int64)
                          for p in self.dataset.idx_per_modality_subset])
                self.dataset._subset_arrays = cache   # the index lists as arrays, built once
            mod_subset_idx = cache[1][idx]
            while len(mod_subset_idx) > 0:
                size = min(len(mod_subset_idx), batch_size)
                (incomplete if size < batch_size else complete).append(batch_idx)
                # == np.random.choice(mod_subset_idx, size, replace=False): the legacy
                # (frozen) RandomState draws permutation(len)[:size] for it
                pick = np.random.permutation(len(mod_subset_idx))[:size]
                new_indices = mod_subset_idx[pick]
                mod_subset_idx = np.delete(mod_subset_idx, pick)
                indices.append(new_indices)
                batch_idx += 1
        complete_order = np.random.choice(complete, size=len(complete), replace=False)
        incomplete_order = np.random.choice(incomplete, size=len(incomplete), replace=False)
        return [indices[i] for i in complete_order] + [indices[i] for i in incomplete_order]

    def shares(self, global_batches):
        """This rank's part of every global batch + its weight in the ranks' mean."""
        W, r = self.world, self.rank
        mine, scales = [], []
        for b in global_batches:
            n = len(b)
            base, rem = divmod(n, W)
            lo = r * base + min(r, rem)
            hi = lo + base + (1 if r < rem else 0)
            if hi > lo:
                mine.append(b[lo:hi])
                scales.append((hi - lo) * W / float(n))
            else:           # fewer samples than ranks: take part with weight 0
                mine.append(b[:1])
                scales.append(0.0)
        return mine, scales

    def __iter__(self):
        if self.world == 1:
            batches = self.draw(self.batch_size)
            self.loss_scales = [1.0] * len(batches)
            return iter(batches)
        from ..parallel import share_schedule
        glob = self.draw(self.batch_size * self.world) if self.rank == 0 else None
        mine, self.loss_scales = self.shares(share_schedule(glob))
        return iter(mine)


def fit_scalers(dataset):
    """{mod: (mean, scale)} of the per-modality StandardScaler the reference fits on the
    training set (experiment.py:146-166: every sample that has the modality, through the
    dataset's __getitem__; population standard deviation, a constant feature keeps
    scale 1) -- float64, straight from the blocks: a block row belongs to exactly one
    subject, so the rows the dataset's subjects point at are the fitted samples."""
    out = {}
    subjects = np.arange(dataset.n_samples) if dataset.indices is None \
        else np.asarray(dataset.indices)
    for mod in dataset.modalities:
        rows = [int(r) for r in dataset.idx_per_mod[mod][subjects] if r is not None]
        x = np.asarray(dataset.data[mod], dtype=np.float64)[rows]
        mean = x.mean(axis=0)
        scale = np.sqrt(((x - mean) ** 2).mean(axis=0))
        scale[scale < 10 * np.finfo(np.float64).eps * np.maximum(1.0, np.abs(mean))] = 1.0
        out[mod] = (mean, scale)
    return out


class ResidentCohort:
    """Per-modality blocks, scaled once and resident in HBM; batches are index
    vectors.  `scalers` = {mod: (mean, scale)} reproduces the per-sample
    StandardScaler.transform of the reference's on-the-fly transform
    (experiment.py:228-232): (x - mean) / scale, in float32."""

    def __init__(self, dataset, device, scalers=None):
        self.dataset = dataset
        self.device = torch.device(device)
        self.x = {}
        for mod in dataset.modalities:
            arr = torch.as_tensor(np.array(dataset.data[mod], dtype=np.float64, order="C"))  # (a copy: the block may be a read-only memory map)
            if scalers is not None and mod in scalers:
                mean, scale = scalers[mod]
                arr = (arr - torch.as_tensor(mean, dtype=torch.float64)) / \
                    torch.as_tensor(scale, dtype=torch.float64)
            self.x[mod] = L.device_rows(arr.to(torch.float32), self.device)
        self._indices = None   # dataset.indices as an array, built on first use
        # block row of every subject, -1 where the modality is missing
        self.rows = {}
        for mod in dataset.modalities:
            col = dataset.idx_per_mod[mod]
            self.rows[mod] = np.array([-1 if r is None else int(r) for r in col], dtype=np.int64)

    def batch(self, sample_indices):
        """(inputs, row_index) for engine.train_step / forward: the modalities
        every sample of the batch has, and their block rows."""
        idx = np.asarray(sample_indices, dtype=np.int64)
        if self.dataset.indices is not None:
            if self._indices is None:
                self._indices = np.ascontiguousarray(self.dataset.indices, dtype=np.int64)
            idx = self._indices[idx]
        inputs, row_index = {}, {}
        for mod in self.dataset.modalities:
            rows = self.rows[mod][idx]
            if (rows >= 0).all():
                inputs[mod] = self.x[mod]
                row_index[mod] = torch.as_tensor(rows, dtype=torch.int32)
            elif (rows >= 0).any():
                raise ValueError("batch mixes samples with and without %r; use "
                                 "MissingModalitySampler" % mod)
        return inputs, row_index

    def epoch(self, batch_size):
        """One epoch of (inputs, row_index) in MissingModalitySampler order; all
        index vectors go to the device in one transfer."""
        out = []
        for i, r, _ in self.epoch_schedule(batch_size):
            out.append((dict(i.x), i.row_index()) if isinstance(i, IndexBatch) else (i, r))
        return out

    # ---------------------------------------------------------------- epoch schedule
    def _staging(self, slot):
        """Pinned int32 staging vectors of one epoch (a row per sample at most), two sets used
        in turn: allocated HERE, by the caller's thread -- the helper thread makes no device
        runtime call at all (its first one would initialise a context of its own, ~100 ms)."""
        pools = self.__dict__.setdefault("_pinned", {})
        if slot not in pools:     # ONE vector for all modalities: one transfer per epoch
            n, M = len(self.dataset), len(self.dataset.modalities)
            flat = torch.empty(n * M, dtype=torch.int32)
            flat = flat.pin_memory() if self.device.type == "cuda" else flat
            pools[slot] = (flat, {m: flat[k * n:(k + 1) * n]
                                  for k, m in enumerate(self.dataset.modalities)})
        return pools[slot]

    def _host_schedule(self, state, batch_size, staging=None):
        """Everything of an epoch that is host work, from a GIVEN RandomState state: the
        sampler's draws and, per modality, the block rows of every batch that holds it as
        ONE pinned int32 vector -- two C calls (the GIL is released meanwhile).  No global
        state is touched, so the helper thread of `epoch_schedule` runs this for the NEXT
        epoch while the GPU is busy with the current one."""
        import ctypes as C
        sampler = MissingModalitySampler(self.dataset, batch_size)
        begin, items = sampler._subset_arrays()
        after, out_items, out_begin, out_subset = sampler.draw_from(state, begin, items,
                                                                    batch_size)
        mods = self.dataset.modalities
        nb, M = len(out_subset), len(mods)
        lens = np.diff(out_begin)
        cache = self.__dict__.setdefault("_sched_cache", {})
        if cache.get("subsets") is not self.dataset.modality_subsets:   # (per cohort, not per epoch)
            cache["subsets"] = self.dataset.modality_subsets
            cache["has"] = np.ascontiguousarray(
                [[m in sub for m in mods] for sub in self.dataset.modality_subsets], dtype=np.uint8)
        has = cache["has"]
        per_batch = has[out_subset].astype(bool) if nb else np.zeros((0, M), bool)
        indices = None
        if self.dataset.indices is not None:
            if self._indices is None:
                self._indices = np.ascontiguousarray(self.dataset.indices, dtype=np.int64)
            indices = self._indices
        rows, starts = {}, {}
        for k, mod in enumerate(mods):
            count = int(lens[per_batch[:, k]].sum()) if nb else 0
            rows[mod] = staging[1][mod][:count] if staging is not None else \
                torch.empty(count, dtype=torch.int32)
            starts[mod] = np.empty(nb, dtype=np.int64)
        ptrs = lambda seq: (C.c_void_p * M)(*seq)
        L.check(L.lib.mopoe_sampler_rows(
            M, nb, out_items.ctypes.data, np.ascontiguousarray(out_begin).ctypes.data,
            np.ascontiguousarray(out_subset).ctypes.data, np.ascontiguousarray(has).ctypes.data,
            indices.ctypes.data if indices is not None else None,
            ptrs(self.rows[m].ctypes.data for m in mods),
            ptrs(rows[m].data_ptr() for m in mods),
            ptrs(starts[m].ctypes.data for m in mods)), "mopoe_sampler_rows")
        return dict(after=after, lens=lens.tolist(), rows=rows, staging=staging,
                    starts={m: v.tolist() for m, v in starts.items()})

    @staticmethod
    def _global_mt():
        """(key, pos) of numpy's global legacy generator IN PLACE -- a view of the 624 state
        words and the position word behind them (numpy's mt19937_state: `uint32_t key[624];
        int pos;`) -- or None where that generator is not the MT19937 the legacy API promises.
        np.random.get_state() + set_state() copy and validate the state: 115 us per epoch,
        a quarter of a 14-step epoch; a look and a write in place are 4 us."""
        import ctypes as C
        try:
            bg = np.random.mtrand._rand._bit_generator
            if type(bg).__name__ != "MT19937":
                return None
            addr = bg.ctypes.state_address
            key = np.ctypeslib.as_array((C.c_uint32 * 624).from_address(addr))
            return key, C.c_int32.from_address(addr + 624 * 4)
        except (AttributeError, TypeError, ValueError):
            return None

    @staticmethod
    def _same_state(a, b):
        return a[0] == b[0] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and \
            np.array_equal(a[1], b[1])

    def epoch_schedule(self, batch_size, world=1, rank=0):
        """One epoch of (inputs, row_index, loss_scale) for rank `rank` of `world`
        data-parallel replicas (MissingModalitySampler's rank-aware schedule).

        One process: `inputs` is an IndexBatch (the resident blocks + the addresses of the
        batch's gather vectors inside ONE device tensor per modality: no per-batch tensor is
        made) and row_index is None.  The epoch's host work is two C calls, and the NEXT
        epoch's is started right away on a helper thread, speculating that nobody draws from
        numpy's global generator in between (the reference's loop does not).  The speculation
        is checked: the prefetched epoch is used only if np.random is exactly in the state it
        was drawn from -- otherwise it is thrown away and the epoch drawn afresh, so the
        global stream always reads as if the reference's sampler had drawn at this moment."""
        if world > 1:
            return self._epoch_schedule_ranks(batch_size, world, rank)
        host = None
        live = self._global_mt()     # (the generator's words in place; None: through get / set_state)
        now = np.random.get_state() if live is None else None
        pre, self._prefetch = getattr(self, "_prefetch", None), None
        if pre is not None:
            try:
                ahead = pre["future"].result()
            except Exception:      # (the same error surfaces from the foreground draw below)
                ahead = None
            if ahead is not None and pre["batch_size"] == batch_size:
                if live is not None:
                    # the stream is where the prefetch started from iff key and position are
                    # (the cached-gaussian words of the legacy state do not enter the draws)
                    same = live[1].value == pre["state"][2] and np.array_equal(live[0], pre["state"][1])
                else:
                    same = self._same_state(pre["state"], now)
                if same:
                    host = ahead
        turn = self.__dict__.get("_turn", 0)
        if host is None:
            if now is None:
                now = np.random.get_state()
            host = self._host_schedule(now, batch_size, self._staging(turn % 2))
        if live is not None:        # the stream moves on by exactly the sampler's draws
            live[0][:] = host["after"][1]
            live[1].value = host["after"][2]
        else:
            np.random.set_state(host["after"])
        if getattr(self, "_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mopoe-sampler")
        mods = self.dataset.modalities
        n_all = len(self.dataset)
        flat = host["staging"][0].to(self.device, non_blocking=True)      # one transfer
        dev = {m: flat[k * n_all:(k + 1) * n_all] for k, m in enumerate(mods)}
        # the OTHER staging set goes to the helper thread; its last copy (the epoch before
        # this one) must have left it
        copied = self.__dict__.setdefault("_copied", {})
        if self.device.type == "cuda":
            if turn % 2 not in copied:
                copied[turn % 2] = torch.cuda.Event()
            copied[turn % 2].record()
            if (turn + 1) % 2 in copied:
                copied[(turn + 1) % 2].synchronize()
        self._turn = turn + 1
        self._prefetch = dict(batch_size=batch_size, state=host["after"],
                              future=self._pool.submit(self._host_schedule, host["after"],
                                                       batch_size, self._staging((turn + 1) % 2)))
        base = {m: dev[m].data_ptr() for m in mods}
        xs = [self.x[m] for m in mods]
        starts = [host["starts"][m] for m in mods]
        keep = (dev, host["rows"])          # (the pinned source outlives the copy)
        out = []
        for k, n in enumerate(host["lens"]):
            x, ptr = {}, {}
            for m, t, st in zip(mods, xs, starts):
                s = st[k]
                if s >= 0:
                    x[m] = t
                    ptr[m] = base[m] + 4 * s
            out.append((IndexBatch(x, n, ptr, keep), None, 1.0))
        return out

    def _epoch_schedule_ranks(self, batch_size, world, rank):
        sampler = MissingModalitySampler(self.dataset, batch_size, world=world, rank=rank)
        batches = list(sampler)
        scales = list(sampler.loss_scales)
        out = []
        flat, spans, start = [], [], 0
        for b in batches:
            inputs, row_index = self.batch(b)
            for mod, r in row_index.items():
                spans.append((len(out), mod, start, len(r)))
                flat.append(r)
                start += len(r)
            out.append((inputs, {}, scales[len(out)]))
        if flat:
            dev = torch.cat(flat).to(self.device, non_blocking=True)
            for bi, mod, start, n in spans:
                out[bi][1][mod] = dev[start:start + n]
        return out

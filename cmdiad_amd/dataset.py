"""Trainer input paths: the feature-to-feature trainer's files + the pair files of the
feature-to-input / input-to-feature heads (FeatureToInputPreTrainTensorDataset, InputToFeaturePreTrainTensorDataset, PairRing below).

On-disk format (written by DoubleRGBPointFeatures with --save_feature_for_fusion, reference
multiple_features.py:815-825 / 942-945; this package's drop-in writes the same): one ``torch.save``d float32 tensor
``[3136, 1536]`` per sample -- columns 0..767 the Point-MAE patch features, 768..1535 the ViT features resized to
56 x 56 -- under ``<root>/train`` and ``<root>/test``.

* ``PreTrainTensorDataset``  -- the reference's dataset class (dataset.py:247-265): same constructor, ``__len__``,
  ``__getitem__`` -> (tensor on the GPU, 0), same file order (``os.listdir``).
* ``FeatureRing``            -- what replaces ``DataLoader(PreTrainTensorDataset, shuffle=True, num_workers=N,
  multiprocessing_context='forkserver')`` (hallucination_network_pretrain.py:216-225) on the critical path: reader
  threads ``torch.load`` the files of the NEXT batches into pinned staging buffers, a copy stream moves each batch into
  a ring of device-resident ``[B, 3136, 1536]`` buffers, and the training loop receives batches that are already in
  HBM (the reference ``torch.load(..., map_location='cuda')``s inside forked workers, one 19.3 MB file at a time).
  With ``resident=True`` (default when the set fits ``resident_limit_bytes``: the ten MVTec 3D-AD classes are 2 650
  samples x 19.3 MB = 51 GB of the 288 GB HBM) every sample is kept in a device-resident ``[n, 3136, 1536]`` cache the
  first time it is read, and later epochs assemble batches by an on-device gather (616 MB per step at ~5 TB/s) --
  disk and the host are off the critical path from epoch 2 on (a training step is 17.6 ms; torch.load alone sustains
  ~2 GB/s, i.e. 320 ms per batch).
  Batch composition and order are the ones the reference's DataLoader would produce under the same global torch seed
  (RandomSampler draws its permutation seed from the global generator; ``drop_last`` as given), so a run is
  reproducible against the reference sample for sample.
"""
import os
import queue
import threading
from pathlib import Path

import torch
from torch.utils.data import Dataset


def _shared_stream(device, role):
    from . import ops          # (lazy: this module is importable without the HIP library; the device ring is not usable without it)
    return ops.shared_stream(device, role)


class PreTrainTensorDataset(Dataset):
    def __init__(self, root_path):
        super().__init__()
        self.root_path = root_path
        self.tensor_paths = os.listdir(self.root_path)

    def __len__(self):
        return len(self.tensor_paths)

    def __getitem__(self, idx):
        tensor = torch.load(Path(self.root_path, self.tensor_paths[idx]), map_location="cuda")
        return tensor, 0


# data_type -> ((attribute stem, sub-directory, glob), (…)): the files of a pair, in the reference's attribute names
_PAIR_LAYOUT = {
    "rgb_fxyz": (("rgb", "rgb", "*.pt"), ("fxyz", "fxyz", "*hfxyz.pt")),
    "xyz_frgb": (("frgb", "frgb", "*.pt"), ("xyz", "xyz", "*.pt")),
}


def _discover_pairs(ds, root_path, data_type):
    """Sets <stem>_root_path / <stem>_paths (sorted as plain strings: bagel10 before bagel2, in both lists alike) and `len` on ds,
    the attributes the reference's classes carry; False for a data_type without a layout."""
    layout = _PAIR_LAYOUT.get(data_type)
    if layout is None:
        return False
    counts = []
    for stem, sub, pattern in layout:
        root = Path(root_path, sub)
        paths = sorted(root.glob(pattern))
        setattr(ds, stem + "_root_path", root)
        setattr(ds, stem + "_paths", paths)
        counts.append(len(paths))
    assert counts[0] == counts[1], f"{root_path}: {counts[0]} / {counts[1]} files of the two kinds"
    ds.len = counts[0]
    return True


class FeatureToInputPreTrainTensorDataset(Dataset):
    """The feature-to-INPUT heads' training pairs (reference dataset.py:268-314; selected at
    hallucination_network_pretrain.py:180-201 for RGBFeatureToXYZInput{MLP,Conv} / XYZFeatureToRGBInput{MLP,Conv}).
    Files as DoubleRGBPointFeatures writes them with --save_frgb_xyz / --save_rgb_fxyz (multiple_features.py:827-867, 947-962;
    this package's drop-in writes the same names):
      data_type 'xyz_frgb': <root>/frgb/<class><i>_frgb.pt [3136, 768]  +  <root>/xyz/<class><i>_xyz.pt [3, 224, 224]  -> (frgb, xyz)
      data_type 'rgb_fxyz': <root>/rgb/<class><i>_rgb.pt [3, 224, 224]  +  <root>/fxyz/<class><i>_hfxyz.pt [3136, 768]  -> (fxyz, rgb)
    (the [784, 768] `_lfxyz.pt` files beside them are not read), both loaded straight onto the GPU, the FEATURE first.  Any other
    data_type leaves the object without a length, as the reference does."""

    device = "cuda"      # torch.load(map_location=...) of __getitem__ (the reference hard-codes 'cuda'; CPU tests override it)

    def __init__(self, root_path, data_type):
        super().__init__()
        self.root_path, self.data_type = root_path, data_type
        _discover_pairs(self, root_path, data_type)

    def __len__(self):
        return self.len

    def pair_paths(self, idx):
        """(first, second) file of sample idx, in the order __getitem__ returns them: feature, then input."""
        if self.data_type == "rgb_fxyz":
            return self.fxyz_paths[idx], self.rgb_paths[idx]
        if self.data_type == "xyz_frgb":
            return self.frgb_paths[idx], self.xyz_paths[idx]
        return None

    def __getitem__(self, idx):
        pair = self.pair_paths(idx)
        if pair is None:
            return None          # (the reference's __getitem__ falls through both branches)
        return tuple(torch.load(f, map_location=self.device) for f in pair)


class InputToFeaturePreTrainTensorDataset(Dataset):
    """The input-to-FEATURE (HRNet) heads' training pairs (reference dataset.py:317-362; selected at
    hallucination_network_pretrain.py:203-214 for RGBInputToXYZFeatureHRNET / XYZInputToRGBFeatureHRNET): the same files as
    FeatureToInputPreTrainTensorDataset in the OTHER order -- 'rgb_fxyz' -> (rgb [3,224,224], fxyz [3136,768]),
    'xyz_frgb' -> (xyz, frgb) -- loaded to the HOST (the reference's DataLoader pins and moves them); any other data_type
    raises NotImplementedError."""

    def __init__(self, root_path, data_type):
        super().__init__()
        self.root_path, self.data_type = root_path, data_type
        if not _discover_pairs(self, root_path, data_type):
            raise NotImplementedError

    def __len__(self):
        return self.len

    def pair_paths(self, idx):
        """(first, second) file of sample idx: input, then feature."""
        if self.data_type == "rgb_fxyz":
            return self.rgb_paths[idx], self.fxyz_paths[idx]
        return self.xyz_paths[idx], self.frgb_paths[idx]

    def __getitem__(self, idx):
        return tuple(torch.load(f) for f in self.pair_paths(idx))


class PairRing:
    """What FeatureRing is for the feature-to-feature trainer, for the PAIR datasets above: one epoch of batches
    ``(first [b, ...], second [b, ...])`` already resident in HBM, in the order and composition the reference's
    ``DataLoader(dataset, shuffle=..., batch_size=..., drop_last=...)`` produces under the same global torch seed
    (`epoch_permutation`).  A pair is 9.6 MB + 0.6 MB, the ten MVTec 3D-AD classes 2 650 pairs = 27 GB: every pair is read from disk
    ONCE (the first epoch, `readers` host threads), kept in two device-resident caches, and every later batch is an on-device
    gather -- no DataLoader workers, no per-step H2D.  Works on the host too (device='cpu': tests)."""

    def __init__(self, dataset, batch_size, shuffle=True, drop_last=True, device="cuda", readers=4):
        self.ds, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.device, self.readers = torch.device(device), readers
        a, b = (torch.load(p, map_location="cpu") for p in dataset.pair_paths(0))
        n = len(dataset)
        self._cache = (torch.empty((n, *a.shape), dtype=a.dtype, device=self.device), torch.empty((n, *b.shape), dtype=b.dtype, device=self.device))
        self._have = torch.zeros(n, dtype=torch.bool)

    def __len__(self):
        n = len(self.ds)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def batches(self):
        order = epoch_permutation(len(self.ds), self.shuffle)
        out = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and out and len(out[-1]) < self.batch_size:
            out.pop()
        return out

    def _fill(self, idxs):
        todo = [i for i in idxs if not bool(self._have[i])]
        if not todo:
            return

        def one(i):
            pa, pb = self.ds.pair_paths(i)
            return i, torch.load(pa, map_location="cpu"), torch.load(pb, map_location="cpu")

        if self.readers > 1 and len(todo) > 1:
            import concurrent.futures as cf
            with cf.ThreadPoolExecutor(max_workers=self.readers) as ex:
                got = list(ex.map(one, todo))
        else:
            got = [one(i) for i in todo]
        for i, a, b in got:
            self._cache[0][i].copy_(a, non_blocking=True)
            self._cache[1][i].copy_(b, non_blocking=True)
            self._have[i] = True

    def __iter__(self):
        for idxs in self.batches():
            self._fill(idxs)
            sel = torch.tensor(idxs, dtype=torch.int64, device=self.device)
            yield self._cache[0].index_select(0, sel), self._cache[1].index_select(0, sel)


def epoch_permutation(n, shuffle):
    """Index order of one epoch exactly as torch's DataLoader produces it: SequentialSampler, or RandomSampler with
    generator=None (a fresh generator seeded from the GLOBAL generator, then randperm)."""
    torch.empty((), dtype=torch.int64).random_()  # the loader iterator's own base seed, drawn first (torch dataloader.py)
    if not shuffle:
        return list(range(n))
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).tolist()


class FeatureRing:
    """Iterable over one epoch of batches ``(features [b, rows, cols] on the device, labels [b] zeros)``.

    depth = device buffers in the ring (>= 2: one being consumed, the others filled ahead); a batch stays valid until
    the next one is requested (its refill is ordered after the work the consumer had queued on it).
    readers = host threads decoding files (torch.load releases the GIL while reading)."""

    def __init__(self, root_path, batch_size, shuffle=True, drop_last=True, device="cuda", depth=3, readers=4,
                 resident=True, resident_limit_bytes=200 << 30):
        assert depth >= 2
        self.root = root_path
        self.files = os.listdir(root_path)
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.device, self.depth, self.readers = torch.device(device), depth, readers
        probe = torch.load(Path(root_path, self.files[0]), map_location="cpu")
        self.sample_shape, self.dtype = tuple(probe.shape), probe.dtype
        pin = self.device.type == "cuda"
        self._staging = [torch.empty((batch_size, *self.sample_shape), dtype=self.dtype, pin_memory=pin) for _ in range(depth)]
        self._dev = [torch.empty((batch_size, *self.sample_shape), dtype=self.dtype, device=self.device) for _ in range(depth)]
        self._copy_stream = _shared_stream(self.device, "dataset.copy") if pin else None
        n_bytes = len(self.files) * probe.numel() * probe.element_size()
        self._cache = None
        self._cached = [False] * len(self.files)
        if resident and pin and n_bytes <= resident_limit_bytes:
            self._cache = torch.empty((len(self.files), *self.sample_shape), dtype=self.dtype, device=self.device)

    def __len__(self):
        n = len(self.files)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def batches(self):
        """The epoch's batches as lists of file indices (consumes the global RNG like the reference's DataLoader)."""
        order = epoch_permutation(len(self.files), self.shuffle)
        out = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and out and len(out[-1]) < self.batch_size:
            out.pop()
        return out

    def __iter__(self):
        plan = self.batches()
        ready = queue.Queue(maxsize=self.depth - 1)     # filled slots waiting for the consumer
        free = queue.Queue()
        for s in range(self.depth):
            free.put(s)
        stop = threading.Event()

        def load_batch(slot, idxs):
            stage = self._staging[slot]
            if self.readers > 1 and len(idxs) > 1:
                def one(j_i):
                    j, i = j_i
                    stage[j].copy_(torch.load(Path(self.root, self.files[i]), map_location="cpu"))
                threads = [threading.Thread(target=one, args=(ji,)) for ji in enumerate(idxs)]
                live = []
                for t in threads:           # at most `readers` files in flight
                    t.start(); live.append(t)
                    if len(live) >= self.readers:
                        live.pop(0).join()
                for t in live:
                    t.join()
            else:
                for j, i in enumerate(idxs):
                    stage[j].copy_(torch.load(Path(self.root, self.files[i]), map_location="cpu"))

        def producer():
            try:
                for idxs in plan:
                    slot = free.get()
                    if stop.is_set():
                        return
                    ev = None
                    if self._cache is not None and all(self._cached[i] for i in idxs):
                        with torch.cuda.stream(self._copy_stream):  # epoch >= 2: on-device gather, no host work
                            sel = torch.tensor(idxs, dtype=torch.int64).to(self.device, non_blocking=True)
                            torch.index_select(self._cache, 0, sel, out=self._dev[slot][:len(idxs)])
                            ev = torch.cuda.Event()
                            ev.record(self._copy_stream)
                        ready.put((slot, len(idxs), ev))
                        continue
                    load_batch(slot, idxs)
                    if self._copy_stream is not None:
                        with torch.cuda.stream(self._copy_stream):
                            self._dev[slot][:len(idxs)].copy_(self._staging[slot][:len(idxs)], non_blocking=True)
                            if self._cache is not None:
                                sel = torch.tensor(idxs, dtype=torch.int64).to(self.device, non_blocking=True)
                                self._cache.index_copy_(0, sel, self._dev[slot][:len(idxs)])
                                for i in idxs:
                                    self._cached[i] = True
                            ev = torch.cuda.Event()
                            ev.record(self._copy_stream)
                            ev.synchronize()  # the pinned staging buffer of this slot is reused for the next disk batch
                    else:
                        self._dev[slot][:len(idxs)].copy_(self._staging[slot][:len(idxs)])
                    ready.put((slot, len(idxs), ev))
                ready.put(None)
            except BaseException as exc:  # surface reader errors in the training loop, not in a dead thread
                ready.put(exc)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        held = []
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                slot, b, ev = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                held.append(slot)
                if len(held) > 1:   # the previous batch's buffer may now be refilled
                    done = held.pop(0)
                    if self._copy_stream is not None:  # ... once the consumer's work queued so far has read it
                        ev2 = torch.cuda.Event()
                        ev2.record(torch.cuda.current_stream(self.device))
                        self._copy_stream.wait_event(ev2)
                    free.put(done)
                yield self._dev[slot][:b], torch.zeros(b, dtype=torch.int64)
        finally:
            stop.set()
            for s in range(self.depth):
                free.put(s)
            th.join(timeout=5)

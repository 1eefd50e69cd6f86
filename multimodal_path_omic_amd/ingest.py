"""Slide feature ingest for window training.

The reference loads one slide per step with a blocking `torch.load` of a (M, 1024) fp32 tensor and a blocking
`.to(device)` (dataset/dataset.py:119-143, models/mcat/main.py:37).  At 1.5 ms of GPU time per 32-slide window that
path -- not the kernels -- bounds a cold epoch, so the feeder

* packs a whole gradient-accumulation window into ONE pinned host slab (slides concatenated along rows, the layout
  `ops.BagBatch` wants) with a pool of loader threads, converting to the bag storage dtype (bf16 halves the bytes on the
  link) while it packs;
* ships the slab with one asynchronous H2D copy on its own stream into a ring of device buffers, `depth` windows ahead
  of the consumer;
* hands the consumer a `BagBatch` whose stream dependency is an event wait -- no host synchronisation anywhere.

Per-slide `.pt` files are read with `torch.load(weights_only=True, mmap=True)` (nothing from the file is executed).
HDF5 (`use_h5_dataset`, dataset/dataset.py:127-129) is not offered: h5py is absent from this image.
"""
from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, List, Optional, Sequence

import torch

from .ops import BagBatch


class SlideStore:
    """Where patch-feature matrices come from: `length(i)` rows of 1024 features for slide i, `load(i)` -> CPU tensor."""

    def __len__(self) -> int:
        raise NotImplementedError

    def length(self, i: int) -> int:
        raise NotImplementedError

    def load(self, i: int) -> torch.Tensor:
        raise NotImplementedError


class ArrayStore(SlideStore):
    """Slides already in host memory (tests, synthetic cohorts)."""

    def __init__(self, slides: Sequence[torch.Tensor]):
        self.slides = list(slides)

    def __len__(self):
        return len(self.slides)

    def length(self, i):
        return int(self.slides[i].shape[0])

    def load(self, i):
        return self.slides[i]


class PtDirStore(SlideStore):
    """The reference's layout: `<patches_dir>/<slide_id>.pt`, one (M, 1024) tensor per file (dataset/dataset.py:124-126).
    Row counts are read once up front (memory-mapped: no data is touched) so windows can be packed before loading."""

    def __init__(self, patches_dir: str, slide_ids: Sequence[str]):
        self.paths = [os.path.join(patches_dir, s.replace(".svs", "") + ("" if s.endswith(".pt") else ".pt")) for s in slide_ids]
        self._len = [int(self._open(p).shape[0]) for p in self.paths]

    @staticmethod
    def _open(path):
        return torch.load(path, map_location="cpu", weights_only=True, mmap=True)

    def __len__(self):
        return len(self.paths)

    def length(self, i):
        return self._len[i]

    def load(self, i):
        t = self._open(self.paths[i])
        if t.dim() == 3:
            t = t.squeeze(0)
        return t


class _Slot:
    def __init__(self, rows, feat, dtype, device):
        self.host = torch.empty(rows, feat, dtype=dtype, pin_memory=True)
        self.dev = torch.empty(rows, feat, dtype=dtype, device=device)
        self.ready = torch.cuda.Event()          # H2D of this slot finished
        self.free = torch.cuda.Event()           # consumer finished with this slot
        self.free_recorded = False
        self.shipped = False                     # `ready` has been recorded at least once


class WindowFeeder:
    """Iterates over windows of `window` slides in `order`, yielding (BagBatch, omics, labels, censorship, ids).

    omics_of(ids) -> list of (B, d_i) CPU tensors; labels_of(ids) / cens_of(ids) -> 1-D CPU tensors.  The returned
    BagBatch aliases a ring slot: it stays valid until `depth` further windows have been requested."""

    def __init__(self, store: SlideStore, order: Sequence[int], window: int, device, omics_of: Callable, labels_of: Callable,
                 cens_of: Callable, bag_dtype=torch.bfloat16, depth: int = 2, workers: int = 8, feat: int = 1024):
        if not torch.cuda.is_available():
            raise RuntimeError("WindowFeeder needs the GPU (pinned host slabs + an H2D copy stream)")
        self.store, self.device, self.dtype, self.feat = store, torch.device(device), bag_dtype, feat
        self.windows = [list(order[i:i + window]) for i in range(0, len(order), window)]
        self.omics_of, self.labels_of, self.cens_of = omics_of, labels_of, cens_of
        cap = max(sum(store.length(i) for i in w) for w in self.windows)
        self.slots = [_Slot(cap, feat, bag_dtype, self.device) for _ in range(depth + 1)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.depth = depth
        self._pending = {}                       # window index -> (slot, {"thread", "lengths" | "error"})

    # ---- producer side
    def _pack(self, w: int, slot: _Slot):
        ids = self.windows[w]
        lengths = [self.store.length(i) for i in ids]
        offs = [0]
        for m in lengths:
            offs.append(offs[-1] + m)

        # The slab is about to be overwritten: the H2D copy of the window it held before must have LEFT it.  That copy
        # waits (on the device) for the consumer of a still older window, so when the GPU lags the host by a ring's worth
        # of windows it may not even have started -- every other ordering here is device-side.  Block THIS packer thread
        # (never the consumer) until the slab's last copy is done.
        if slot.shipped:
            slot.ready.synchronize()

        def one(k):
            src = self.store.load(ids[k])
            if src.shape != (lengths[k], self.feat):
                raise ValueError(f"slide {ids[k]}: expected {(lengths[k], self.feat)}, got {tuple(src.shape)}")
            slot.host[offs[k]:offs[k + 1]].copy_(src)          # converts to the bag dtype while packing
        list(self.pool.map(one, range(len(ids))))
        with torch.cuda.stream(self.copy_stream):
            if slot.free_recorded:
                self.copy_stream.wait_event(slot.free)         # the consumer must be done with the slot's old window
            slot.dev[:offs[-1]].copy_(slot.host[:offs[-1]], non_blocking=True)
            slot.ready.record(self.copy_stream)
            slot.shipped = True
        return lengths

    def _submit(self, w: int):
        """Start packing + shipping window w on a driver thread (the per-slide loads fan out over the pool)."""
        if w >= len(self.windows) or w in self._pending:
            return
        slot = self.slots[w % len(self.slots)]
        holder = {}

        def run():
            try:
                holder["lengths"] = self._pack(w, slot)
            except BaseException as e:          # surfaced to the consumer in __iter__
                holder["error"] = e
        holder["thread"] = threading.Thread(target=run, daemon=True)
        holder["thread"].start()
        self._pending[w] = (slot, holder)

    # ---- consumer side
    def __len__(self):
        return len(self.windows)

    def __iter__(self):
        for w in range(min(self.depth, len(self.windows))):
            self._submit(w)
        prev_slot: Optional[_Slot] = None
        for w in range(len(self.windows)):
            slot, holder = self._pending.pop(w)
            holder["thread"].join()
            if "error" in holder:
                raise holder["error"]
            cur = torch.cuda.current_stream(self.device)
            if prev_slot is not None:                          # everything enqueued so far has used the previous window
                prev_slot.free.record(cur)
                prev_slot.free_recorded = True
            self._submit(w + self.depth)
            cur.wait_event(slot.ready)
            lengths = holder["lengths"]
            ids = self.windows[w]
            total = sum(lengths)
            bags = BagBatch.from_lengths(slot.dev[:total], lengths)
            omics = [o.to(self.device, non_blocking=True) for o in self.omics_of(ids)]
            labels = self.labels_of(ids).to(self.device, non_blocking=True)
            cens = self.cens_of(ids).to(self.device, non_blocking=True)
            prev_slot = slot
            yield bags, omics, labels, cens, ids

    def close(self):
        self.pool.shutdown(wait=True)

"""The step before the hot path: the reference's on-disk feature format, its dataset interface, and an MI355X-first batch
pipeline.

Format (reference: util/preprocessing/data_writer.py:11-38,71-83; file names datagroup.py:203, preprocess_data.py:57-67):
``<modality>_<split>_features.npy`` is a NumPy v1.0 file whose header occupies exactly 128 bytes (the writer memory-maps
the raw array at offset 128 first and writes the header over the gap afterwards) followed by the raw little-endian array
``(num_samples, M, T, V, C)``; ``<split>_labels.npy`` holds the integer labels.  ``np.load(path, mmap_mode="r")`` reads it.

Reader (reference: torch_src/loader.py:21-33, torch_src/dataset.py:11-58): ``NumpyDatasetLoader`` / ``MultiModalDataset`` with
the same names and semantics (feature id = file-name prefix before the first underscore; one modality -> array, several ->
dict).

Batches (reference: ``DataLoader(dataset, batch_size, shuffle, drop_last)`` + a synchronous unpinned
``features_batch.float().cuda()`` per step, session/training.py:21-26, session/session.py:168-174): ``ClipBatches`` yields
the same ``(features, labels, indices)`` triples, already on the device, in one of two ways
  * resident: the whole split is uploaded once (NTU-RGB-D cross-subject train: 7.2 GB of 288 GB) and a batch is a device-side
    row gather -- no PCIe traffic per step at all;
  * streaming: rows are gathered from the memory map into one of two pinned host buffers and copied on a side HIP stream
    while the previous step computes (double buffering; the consumer waits on an event, never on the host).
Data-parallel ranks take contiguous shards of every global batch (``dp.shard_batch``), all ranks shuffling with the same
seed.  torch is plumbing here (pinned memory, streams, index_select); nothing on this path calls the HIP kernels.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, Optional, Sequence, Tuple, Union

import numpy as np
import numpy.lib.format
import torch

from .dp import shard_batch

HEADER_BYTES = 128      # data_writer.py:19: np.memmap(out_path, dtype, "w+", 128, shape)


# ---- writer (util/preprocessing/data_writer.py) -------------------------------------------------------------------------
class MemoryMappedArray:
    """Raw array memory-mapped at byte 128 of ``out_path``; the v1.0 ``.npy`` header is written over the gap on close."""

    def __init__(self, out_path: str, dtype: type, shape: Sequence[int]):
        self.out_path = out_path
        self.dtype = dtype
        self.shape = tuple(shape)
        self.data = None

    def create_file(self):
        if self.out_path:
            self.data = np.memmap(self.out_path, self.dtype, "w+", HEADER_BYTES, self.shape)

    def close_file(self):
        if self.data is not None:
            self.data.flush()
            MemoryMappedArray._write_header(self.data, self.out_path)
        self.data = None

    def __enter__(self):
        self.create_file()
        return self.data

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close_file()

    @staticmethod
    def _write_header(data: np.ndarray, out_path: str):
        header = np.lib.format.header_data_from_array_1_0(data)
        with open(out_path, "r+b") as file:
            np.lib.format.write_array_header_1_0(file, header)
            if file.tell() != HEADER_BYTES:     # the reference silently corrupts the first samples in this case
                raise ValueError(f"{out_path}: the .npy header takes {file.tell()} bytes, the format reserves {HEADER_BYTES}")


class NumpyWriter:
    """``with NumpyWriter(path, np.float32, (num_samples, M, T, V, C)) as w: w.collect_next(sample)``."""

    def __init__(self, out_path: str, dtype: type, shape: Sequence[int]):
        self.out_path = out_path
        self.sample_index = 0
        self._data_store = MemoryMappedArray(out_path, dtype, shape)

    def start_collect(self):
        self._data_store.create_file()

    def end_collect(self):
        self._data_store.close_file()

    def collect_next(self, sequence, sample_index: int = None):
        sample_index = sample_index or self.sample_index       # (the reference's `or`: an explicit index 0 means "next")
        self._data_store.data[sample_index] = sequence
        self.sample_index += 1

    def __enter__(self):
        self.start_collect()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.end_collect()


# ---- reader (torch_src/loader.py, torch_src/dataset.py) ---------------------------------------------------------------------
class NumpyDatasetLoader:
    def __init__(self, **kwargs):
        self._mmap_mode = None if kwargs.get("in_memory", False) else "r"

    def load_data(self, path: str):
        return np.load(path, self._mmap_mode)

    def index_data_sample(self, data: np.ndarray, index: int) -> np.ndarray:
        return np.array(data[index])

    def get_sample_shape(self, data: np.ndarray) -> Sequence[int]:
        return data.shape[1:]


class MultiModalDataset(torch.utils.data.Dataset):
    """Load data from multiple paths each using their own loader (same interface as the reference's class)."""

    def __init__(self, input_data: Sequence[Tuple[str, NumpyDatasetLoader]], split: str, debug=False):
        assert len(input_data) > 0, "Must at least specify one data path"
        self.labels_data = np.load(os.path.join(input_data[0][0], f"{split}_labels.npy"))
        self.features_data: Dict[str, tuple] = {}
        for input_path, input_loader in input_data:
            for file in filter(lambda f: "features" in f.name and split in f.name and f.is_file(),
                               sorted(os.scandir(input_path), key=lambda f: f.name)):
                feature_id = file.name[:file.name.index("_")]
                self.features_data[feature_id] = (input_loader, input_loader.load_data(file.path))
        if debug:
            self.labels_data = self.labels_data[:100]

    def __len__(self):
        return len(self.labels_data)

    def __getitem__(self, index: int):
        if len(self.features_data) == 1:
            loader, data = next(iter(self.features_data.values()))
            features = loader.index_data_sample(data, index)
        else:
            features = {k: loader.index_data_sample(data, index) for k, (loader, data) in self.features_data.items()}
        return features, self.labels_data[index], index

    def get_input_shape(self) -> dict:
        return {k: loader.get_sample_shape(data) for k, (loader, data) in self.features_data.items()}

    def get_num_classes(self) -> int:
        return len(np.unique(self.labels_data))


# ---- batches ------------------------------------------------------------------------------------------------------------------
Features = Union[torch.Tensor, Dict[str, torch.Tensor]]


class ClipBatches:
    """Iterate one epoch of ``(features, labels, indices)`` device batches over a ``MultiModalDataset``.

    ``batch_size`` is the global batch (the reference's config value); rank r of ``world`` receives clips
    ``[r*bs/world, (r+1)*bs/world)`` of each global batch.  ``shuffle`` permutes with ``seed + epoch`` (``set_epoch``), the
    same permutation on every rank.  ``resident``: True / False, or None = upload the split when it takes at most
    ``resident_budget`` bytes.  Features come out float32, labels int64 (session.py:171-174)."""

    def __init__(self, dataset: MultiModalDataset, batch_size: int, *, shuffle: bool = False, drop_last: bool = False,
                 seed: int = 1, rank: int = 0, world: int = 1, device: Union[str, torch.device] = "cuda",
                 resident: Optional[bool] = None, resident_budget: int = 64 << 30):
        if batch_size % world:
            raise ValueError(f"global batch {batch_size} is not divisible by world size {world}")
        self.ds, self.bs, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.seed, self.rank, self.world, self.epoch = seed, rank, world, 0
        self.device = torch.device(device)
        self.keys = list(dataset.features_data)
        self.arrays = {k: dataset.features_data[k][1] for k in self.keys}
        n = len(dataset)
        for k, a in self.arrays.items():
            if not isinstance(a, np.ndarray) or len(a) < n:
                raise TypeError(f"ClipBatches needs array-backed features with >= {n} samples (feature {k!r})")
        nbytes = sum(int(np.prod(a.shape[1:])) * 4 * n for a in self.arrays.values())
        self.resident = nbytes <= resident_budget if resident is None else bool(resident)
        self.labels = torch.from_numpy(np.asarray(dataset.labels_data).astype(np.int64))
        on_gpu = self.device.type == "cuda"
        if self.resident:
            self.dev_feat = {k: torch.from_numpy(np.array(a[:n], dtype=np.float32)).to(self.device)
                             for k, a in self.arrays.items()}
            self.dev_labels = self.labels.to(self.device)
        else:
            per = batch_size // world
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=on_gpu)      # noqa: E731
            self.host = [{k: mk((per, *a.shape[1:]), torch.float32) for k, a in self.arrays.items()} for _ in range(2)]
            self.host_lab = [mk((per,), torch.int64) for _ in range(2)]
            self.dev = [{k: torch.empty((per, *a.shape[1:]), dtype=torch.float32, device=self.device)
                         for k, a in self.arrays.items()} for _ in range(2)]
            self.dev_lab = [torch.empty((per,), dtype=torch.int64, device=self.device) for _ in range(2)]
            self.copy_stream = torch.cuda.Stream(self.device) if on_gpu else None
            self.copied = [torch.cuda.Event() if on_gpu else None for _ in range(2)]
            self.consumed = [torch.cuda.Event() if on_gpu else None for _ in range(2)]

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        """Batches per epoch, the same on every rank (a ragged tail shorter than the world size yields no batch)."""
        n = len(self.ds)
        full, tail = divmod(n, self.bs)
        return full + int(not self.drop_last and tail >= self.world)

    def _order(self) -> torch.Tensor:
        n = len(self.ds)
        if not self.shuffle:
            return torch.arange(n)
        return torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + self.epoch))

    def _shards(self) -> Iterator[torch.Tensor]:
        order = self._order()
        for b in range((len(order) + self.bs - 1) // self.bs if not self.drop_last else len(order) // self.bs):
            glob = order[b * self.bs:(b + 1) * self.bs]
            if len(glob) != self.bs:
                # ragged last batch (drop_last=False).  Every rank must step the same number of times with the same shard
                # size: ranks average their gradients with equal weight in ONE collective, so an empty or shorter shard would
                # hang the all-reduce or skew the mean.  The tail is trimmed to a multiple of the world size (at most
                # world - 1 clips of the epoch are left out, none when world == 1) and skipped when nothing is left.
                glob = glob[:len(glob) // self.world * self.world]
                if len(glob) == 0:
                    continue
            yield glob[shard_batch(len(glob), self.rank, self.world)]

    def _out(self, feats: Dict[str, torch.Tensor]) -> Features:
        return feats[self.keys[0]] if len(self.keys) == 1 else feats

    def __iter__(self) -> Iterator[Tuple[Features, torch.Tensor, torch.Tensor]]:
        if self.resident:
            for idx in self._shards():
                di = idx.to(self.device)
                yield self._out({k: v.index_select(0, di) for k, v in self.dev_feat.items()}), self.dev_labels.index_select(0, di), idx
            return
        on_gpu = self.device.type == "cuda"

        def stage(slot: int, idx: torch.Tensor) -> int:
            """Gather rows ``idx`` into pinned slot ``slot`` and start their copy to the device on the copy stream."""
            m = len(idx)
            if on_gpu:
                self.copied[slot].synchronize()          # the slot's previous H2D copy has left the pinned buffer
            order = np.sort(idx.numpy())                 # ascending file offsets for the memory map; undone below
            back = np.argsort(np.argsort(idx.numpy()))
            for k, a in self.arrays.items():
                rows = np.asarray(a[order], dtype=np.float32)
                self.host[slot][k][:m].copy_(torch.from_numpy(rows[back]))
            self.host_lab[slot][:m].copy_(self.labels[idx])
            if on_gpu:
                self.copy_stream.wait_event(self.consumed[slot])      # the consumer of the device slot's last batch is done
                with torch.cuda.stream(self.copy_stream):
                    for k in self.keys:
                        self.dev[slot][k][:m].copy_(self.host[slot][k][:m], non_blocking=True)
                    self.dev_lab[slot][:m].copy_(self.host_lab[slot][:m], non_blocking=True)
                    self.copied[slot].record(self.copy_stream)
            else:
                for k in self.keys:
                    self.dev[slot][k][:m].copy_(self.host[slot][k][:m])
                self.dev_lab[slot][:m].copy_(self.host_lab[slot][:m])
            return m

        shards = list(self._shards())
        if on_gpu:
            for e in self.copied + self.consumed:
                e.record(torch.cuda.current_stream(self.device))
        pending = stage(0, shards[0]) if shards else 0
        for i, idx in enumerate(shards):
            slot, m = i % 2, pending
            if i + 1 < len(shards):
                pending = stage((i + 1) % 2, shards[i + 1])           # overlaps the step that consumes batch i
            if on_gpu:
                torch.cuda.current_stream(self.device).wait_event(self.copied[slot])
            feats = {k: self.dev[slot][k][:m] for k in self.keys}
            yield self._out(feats), self.dev_lab[slot][:m], idx
            if on_gpu:
                self.consumed[slot].record(torch.cuda.current_stream(self.device))

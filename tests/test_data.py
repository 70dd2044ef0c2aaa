"""Row f2 of SURVEY.md section 8: the reference's on-disk feature format, dataset interface and the batch pipeline.

Format pins: the reference's reader is ``np.load(path, mmap_mode="r")`` (torch_src/loader.py:25-26) and its writer maps the
raw array at byte 128 (util/preprocessing/data_writer.py:19) -- a file written here must satisfy both, byte for byte."""
import os

import numpy as np
import pytest
import torch


def write_split(root, split, n, shapes, classes=5, seed=0):
    from fusion_gcn_amd.data import NumpyWriter
    rng = np.random.default_rng(seed)
    arrays = {}
    for name, shape in shapes.items():
        a = rng.standard_normal((n, *shape)).astype(np.float32)
        with NumpyWriter(os.path.join(root, f"{name}_{split}_features.npy"), np.float32, a.shape) as w:
            for s in a:
                w.collect_next(s)
        arrays[name] = a
    labels = rng.integers(0, classes, n)
    np.save(os.path.join(root, f"{split}_labels.npy"), labels)
    return arrays, labels


def test_feature_file_format(tmp_path):
    from fusion_gcn_amd.data import HEADER_BYTES, MemoryMappedArray, NumpyDatasetLoader
    arrays, _ = write_split(str(tmp_path), "train", 7, {"skeleton": (2, 10, 25, 3)})
    path = tmp_path / "skeleton_train_features.npy"
    raw = path.read_bytes()
    assert raw[:6] == b"\x93NUMPY" and raw[6:8] == b"\x01\x00"                 # v1.0
    assert int.from_bytes(raw[8:10], "little") + 10 == HEADER_BYTES            # header fills exactly the reserved 128 bytes
    assert len(raw) == HEADER_BYTES + arrays["skeleton"].nbytes
    assert raw[HEADER_BYTES:] == arrays["skeleton"].tobytes()                  # raw C-order float32 right behind it
    got = np.load(path, mmap_mode="r")                                         # the reference's reader
    assert isinstance(got, np.memmap) and got.offset == HEADER_BYTES and got.dtype == np.float32
    assert np.array_equal(got, arrays["skeleton"])
    loader = NumpyDatasetLoader()
    data = loader.load_data(str(path))
    assert tuple(loader.get_sample_shape(data)) == (2, 10, 25, 3)
    assert np.array_equal(loader.index_data_sample(data, 3), arrays["skeleton"][3])
    # a file written by plain numpy (header also 128 bytes for these shapes) reads the same way
    np.save(tmp_path / "plain.npy", arrays["skeleton"])
    assert np.array_equal(loader.load_data(str(tmp_path / "plain.npy")), arrays["skeleton"])
    # a header that cannot fit the reserved 128 bytes is refused instead of overwriting the first sample
    with pytest.raises(ValueError):
        with MemoryMappedArray(str(tmp_path / "big.npy"), np.float32, (1,) * 40):
            pass


def test_multimodal_dataset_interface(tmp_path):
    from fusion_gcn_amd.data import MultiModalDataset, NumpyDatasetLoader
    arrays, labels = write_split(str(tmp_path), "train", 9, {"skeleton": (1, 6, 20, 3), "inertial": (6, 6)})
    write_split(str(tmp_path), "val", 4, {"skeleton": (1, 6, 20, 3), "inertial": (6, 6)}, seed=1)
    ds = MultiModalDataset([(str(tmp_path), NumpyDatasetLoader())], "train")
    assert len(ds) == 9 and set(ds.features_data) == {"skeleton", "inertial"}
    assert {k: tuple(v) for k, v in ds.get_input_shape().items()} == {"skeleton": (1, 6, 20, 3), "inertial": (6, 6)}
    assert ds.get_num_classes() == len(np.unique(labels))
    f, lab, idx = ds[5]
    assert idx == 5 and lab == labels[5] and np.array_equal(f["inertial"], arrays["inertial"][5])
    only = tmp_path / "one"
    only.mkdir()
    a1, l1 = write_split(str(only), "val", 3, {"skeleton": (2, 4, 18, 2)})
    ds1 = MultiModalDataset([(str(only), NumpyDatasetLoader(in_memory=True))], "val")
    f, lab, idx = ds1[2]
    assert isinstance(f, np.ndarray) and np.array_equal(f, a1["skeleton"][2])          # one modality: plain array


@pytest.mark.parametrize("resident", [True, False])
def test_clip_batches_cover_an_epoch_once_and_shard_by_rank(tmp_path, resident):
    from fusion_gcn_amd.data import ClipBatches, MultiModalDataset, NumpyDatasetLoader
    arrays, labels = write_split(str(tmp_path), "train", 22, {"skeleton": (2, 5, 25, 3)})
    ds = MultiModalDataset([(str(tmp_path), NumpyDatasetLoader())], "train")
    seen = []
    for rank in range(2):
        it = ClipBatches(ds, 8, shuffle=True, drop_last=False, seed=1, rank=rank, world=2, device="cpu", resident=resident)
        it.set_epoch(3)
        assert len(it) == 3
        for feats, lab, idx in it:
            assert feats.dtype == torch.float32 and lab.dtype == torch.int64
            assert np.array_equal(feats.numpy(), arrays["skeleton"][idx.numpy()])
            assert np.array_equal(lab.numpy(), labels[idx.numpy()])
            seen.append(idx.clone())
    allidx = torch.cat(seen)
    assert sorted(allidx.tolist()) == list(range(22))                               # every clip exactly once per epoch
    # rank shards of one global batch are the two halves of the same permutation slice
    order = torch.randperm(22, generator=torch.Generator().manual_seed(1 + 3))
    assert torch.equal(seen[0], order[:4]) and torch.equal(seen[3], order[4:8])
    # drop_last as the reference's training DataLoader; another epoch reshuffles
    it = ClipBatches(ds, 8, shuffle=True, drop_last=True, seed=1, device="cpu", resident=resident)
    assert len(it) == 2 and sum(len(i) for _, _, i in it) == 16
    first = [i.clone() for _, _, i in it]
    it.set_epoch(1)
    assert not torch.equal(first[0], next(iter(it))[2])
    it_ns = ClipBatches(ds, 8, shuffle=False, device="cpu", resident=resident)
    assert torch.equal(next(iter(it_ns))[2], torch.arange(8))


@pytest.mark.gpu
@pytest.mark.parametrize("resident", [True, False])
def test_clip_batches_on_the_device(tmp_path, resident):
    """Pinned double-buffered streaming and the HBM-resident gather hand the model the same batches."""
    from fusion_gcn_amd.data import ClipBatches, MultiModalDataset, NumpyDatasetLoader
    arrays, labels = write_split(str(tmp_path), "train", 50, {"skeleton": (2, 30, 25, 3), "inertial": (30, 6)})
    ds = MultiModalDataset([(str(tmp_path), NumpyDatasetLoader())], "train")
    it = ClipBatches(ds, 16, shuffle=True, seed=1, device="cuda:0", resident=resident)
    total = 0
    keep = []
    for feats, lab, idx in it:
        assert feats["skeleton"].is_cuda and lab.is_cuda
        keep.append((feats["skeleton"].sum(), idx))        # consume asynchronously, check after the loop
        assert torch.equal(feats["inertial"].cpu(), torch.from_numpy(arrays["inertial"][idx.numpy()]))
        assert torch.equal(lab.cpu(), torch.from_numpy(labels[idx.numpy()].astype(np.int64)))
        total += len(idx)
    assert total == 50
    for s, idx in keep:
        want = torch.from_numpy(arrays["skeleton"][idx.numpy()]).double().sum()
        assert abs(float(s) - float(want)) < 1e-2

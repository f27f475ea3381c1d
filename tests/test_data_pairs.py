"""utils.data_partial.VAL_DATASET_BYTE (the verification pair sets of the reference, utils/data_partial.py:63-92): pairing, the one
shuffle from Python's `random`, labels, resize on access -- against the reference's recipe restated inline on a small array."""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def test_pair_dataset_follows_the_reference_recipe(tmp_path):
    from utils.data_partial import VAL_DATASET_BYTE
    rng = np.random.default_rng(3)
    imgs = rng.standard_normal((20, 3, 8, 8)).astype(np.float32)
    flags = rng.integers(0, 2, 10).astype(bool)
    stem = str(tmp_path / "lfw")
    np.save(stem + ".npy", imgs)
    np.save(stem + "_list.npy", flags)
    conf = types.SimpleNamespace(img_size=8)
    random.seed(11)
    ds = VAL_DATASET_BYTE(stem, conf)
    after = random.random()
    # the reference: reshape to pairs, ONE random.shuffle of range(len), index pairs and labels with it
    random.seed(11)
    permute = list(range(10))
    random.shuffle(permute)
    assert after == random.random()                       # same consumption of the `random` stream
    pairs = imgs.reshape(10, 2, 3, 8, 8)[permute]
    assert len(ds) == 10
    for i in (0, 3, 9):
        pair, lab = ds[i]
        assert isinstance(pair, torch.Tensor) and pair.dtype == torch.float32 and tuple(pair.shape) == (2, 3, 8, 8)
        assert np.array_equal(pair.numpy(), pairs[i]) and lab == flags[permute][i]
    # another input size: bilinear resize of both images of the pair
    ds2 = VAL_DATASET_BYTE(None, types.SimpleNamespace(img_size=16), images=imgs, labels=flags)
    pair, _ = ds2[1]
    assert tuple(pair.shape) == (2, 3, 16, 16) and torch.isfinite(pair).all()
    # batches collate to [B, 2, C, H, W] as Model._shared_eval_step expects ('b p c h w')
    batch, labs = next(iter(torch.utils.data.DataLoader(ds, batch_size=4)))
    assert tuple(batch.shape) == (4, 2, 3, 8, 8) and labs.shape[0] == 4

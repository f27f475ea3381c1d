"""Verification pair sets: the part of the reference's `utils/data_partial.py` that feeds `Model.validation_step` /
`test_step` -- `VAL_DATASET_BYTE` (/root/reference/utils/data_partial.py:63-92).

The reference memory-maps a bcolz carray `<data_dir>` of N images [N, C, H, W] plus `<data_dir>_list.npy` with N/2 same / different
flags, pairs the images up ([N/2, 2, C, H, W]), shuffles pairs and flags with one permutation from Python's `random`, and resizes every
pair to the model's input size on access.  bcolz is not available here (and is unmaintained), so the array side reads plain `.npy`
(`<data_dir>.npy`, memory-mapped: a 12 000-image LFW set is 450 MB, the large sets several GB) or takes an array; everything else is the
reference's behaviour: same constructor, same `__len__` / `__getitem__` results, same consumption of the `random` stream, so a set
exported with `numpy.save(data_dir + '.npy', numpy.asarray(bcolz_carray))` is read in the reference's order.  The training set
(`CustomImageFolder` + albumentations) stays out of scope; its device-side transform chain is `utils/device_transform.py`.
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset


class VAL_DATASET_BYTE(Dataset):
    def __init__(self, data_dir, conf=None, images=None, labels=None):
        """data_dir: path stem (`<data_dir>.npy` images, `<data_dir>_list.npy` flags); or pass the arrays themselves"""
        super().__init__()
        if images is None:
            path = data_dir if str(data_dir).endswith(".npy") else str(data_dir) + ".npy"
            if not os.path.exists(path):
                raise FileNotFoundError("%s: export the reference's bcolz directory with numpy.save first (bcolz is not available here)" % path)
            images = np.load(path, mmap_mode="r")
            data_dir = str(data_dir)[:-4] if str(data_dir).endswith(".npy") else data_dir
        if labels is None:
            labels = np.load("%s_list.npy" % data_dir)
        n, c, h, w = np.shape(images)
        self.pair_arr = np.reshape(images, [n // 2, 2, c, h, w])              # a view: memory-mapped sets stay on disk
        self.label_arr = np.asarray(labels)
        assert np.shape(self.pair_arr)[0] == np.shape(self.label_arr)[0], "Not match size of patch and label !!!"
        permute = list(range(len(self.label_arr)))
        random.shuffle(permute)                                                # the reference's one draw from `random` (:74-75)
        self._permute = np.asarray(permute, dtype=np.int64)                    # applied on access instead of copying the set
        self.label_arr = self.label_arr[self._permute]
        self.conf = conf

    def __len__(self):
        return len(self.label_arr)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        pair = torch.Tensor(np.ascontiguousarray(self.pair_arr[self._permute[idx]]))
        size = getattr(self.conf, "img_size", None)
        if size is not None and tuple(pair.shape[-2:]) != (size, size):
            # torchvision's Resize((s, s)) on a float tensor = bilinear interpolation with antialiasing (:87)
            lead = pair.shape[:-3]
            pair = torch.nn.functional.interpolate(pair.reshape((-1,) + tuple(pair.shape[-3:])), size=(size, size), mode="bilinear",
                                                   align_corners=False, antialias=True).reshape(tuple(lead) + (pair.shape[-3], size, size))
        return pair, self.label_arr[idx]

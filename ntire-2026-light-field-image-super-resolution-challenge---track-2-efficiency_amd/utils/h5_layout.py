"""N4: the on-disk layout of the test scenes.  The reference writes each scene as an HDF5 file whose datasets are the
TRANSPOSED mosaics (Generate_Data_for_Test.py:88-92: `Lr_SAI_y.transpose((1, 0))`, `Sr_SAI_cbcr.transpose((2, 1, 0))`, `Hr_SAI_y.transpose((1, 0))`,
dtype single) and undoes that at load time (utils/utils_datasets.py:111-128), then `ToTensor()`s the arrays.  This module holds
the layout conversion both ways on plain arrays -- what a loader needs around the h5py calls -- and a loader that uses h5py when
the environment has it (this image does not)."""
import numpy as np
import torch


def from_h5_arrays(Lr_SAI_y, Hr_SAI_y, Sr_SAI_cbcr):
    """arrays as stored in the file -> (Lr (1, A h, A w), Hr (1, A h s, A w s), cbcr (C, A h s, A w s)) float32 tensors,
    following utils/utils_datasets.py:116-128 incl. its degenerate-cbcr cases (missing -> two zero channels, 2-D -> one channel)"""
    lr = np.transpose(np.asarray(Lr_SAI_y), (1, 0))
    hr = np.transpose(np.asarray(Hr_SAI_y), (1, 0))
    cc = np.asarray(Sr_SAI_cbcr, dtype="single") if Sr_SAI_cbcr is not None else np.zeros((), dtype="single")
    if cc.ndim == 3:
        cc = np.transpose(cc, (2, 1, 0))
    elif cc.ndim == 0 or cc.size == 0:
        cc = np.zeros((hr.shape[0], hr.shape[1], 2), dtype=np.float32)
    elif cc.ndim == 2:
        cc = np.expand_dims(cc, axis=-1)
    to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return to_t(lr)[None], to_t(hr)[None], to_t(cc).permute(2, 0, 1).contiguous()   # ToTensor(): (H, W[, C]) -> (C, H, W)


def to_h5_arrays(Lr_SAI_y, Hr_SAI_y, Sr_SAI_cbcr):
    """(A h, A w), (A h s, A w s), (A h s, A w s, 2) mosaics -> the arrays Generate_Data_for_Test.py:88-92 stores (single precision)"""
    f = lambda a, axes: np.ascontiguousarray(np.transpose(np.asarray(a, dtype="single"), axes))
    return f(Lr_SAI_y, (1, 0)), f(Hr_SAI_y, (1, 0)), f(Sr_SAI_cbcr, (2, 1, 0))


def load_test_scene(path):
    """one scene file -> the three tensors of TestSetDataLoader.__getitem__ (needs h5py)"""
    try:
        import h5py
    except ImportError as e:   # pragma: no cover - this image has no h5py
        raise RuntimeError("reading .h5 scenes needs h5py, which this environment does not provide") from e
    with h5py.File(path, "r") as hf:
        return from_h5_arrays(np.array(hf.get("Lr_SAI_y")), np.array(hf.get("Hr_SAI_y")), np.array(hf.get("Sr_SAI_cbcr"), dtype="single"))

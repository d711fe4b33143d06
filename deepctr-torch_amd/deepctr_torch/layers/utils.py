"""Small helpers with the reference's names (layers/utils.py:12-70)."""
import numpy as np
import torch


def concat_fun(inputs, axis=-1):
    """``torch.cat`` that leaves a single tensor untouched (reference layers/utils.py:12-16)."""
    return inputs[0] if len(inputs) == 1 else torch.cat(inputs, dim=axis)


def slice_arrays(arrays, start=None, stop=None):
    """Keras-style slicing of one array or a list of arrays, by range or by an index list
    (reference layers/utils.py:19-70; used by ``fit(validation_split=...)``)."""
    if arrays is None:
        return [None]
    if isinstance(arrays, np.ndarray):
        arrays = [arrays]
    by_index = hasattr(start, '__len__')
    if by_index and stop is not None and isinstance(start, list):
        raise ValueError('The stop argument has to be None if the value of start is a list.')
    if by_index and hasattr(start, 'shape'):
        start = start.tolist()

    def cut(x):
        if x is None:
            return None
        return x[start] if by_index else x[start:stop]

    if isinstance(arrays, list):
        if not by_index and len(arrays) == 1:
            return arrays[0][start:stop]
        return [cut(x) for x in arrays]
    if by_index or hasattr(start, '__getitem__'):
        return cut(arrays)
    return [None]

# -*- coding: utf-8 -*-
"""Feature-column schema (API kept verbatim from the reference, ``deepctr_torch/inputs.py:20-245``)
plus the helpers every model calls.  What differs is underneath: the per-feature ``nn.Embedding``
calls, masks, pooling layers and ``torch.cat`` chains of the reference are replaced by one fused
gfx950 kernel (``csrc/embed.hip``) reached through :mod:`deepctr_torch._hip`.
"""
from collections import OrderedDict, defaultdict, namedtuple
from itertools import chain

import torch
import torch.nn as nn

DEFAULT_GROUP_NAME = "default_group"


# --------------------------------------------------------------------------------------------------
# the three column types (reference inputs.py:20-87): immutable tuples hashed by feature name
# --------------------------------------------------------------------------------------------------
_SPARSE_FIELDS = ('name', 'vocabulary_size', 'embedding_dim', 'use_hash', 'dtype', 'embedding_name',
                  'group_name')


class SparseFeat(namedtuple('SparseFeat', _SPARSE_FIELDS)):
    """A categorical feature stored as one id column of X (reference inputs.py:20-38)."""
    __slots__ = ()

    def __new__(cls, name, vocabulary_size, embedding_dim=4, use_hash=False, dtype="int32",
                embedding_name=None, group_name=DEFAULT_GROUP_NAME):
        if embedding_dim == "auto":  # reference inputs.py:29-30
            embedding_dim = 6 * int(pow(vocabulary_size, 0.25))
        if use_hash:  # accepted and ignored, exactly like the reference (inputs.py:31-33)
            print("Notice! Feature Hashing on the fly currently is not supported in torch version,"
                  "you can use tensorflow version!")
        return super().__new__(cls, name, vocabulary_size, embedding_dim, use_hash, dtype,
                               name if embedding_name is None else embedding_name, group_name)

    def __hash__(self):
        return hash(self.name)


class VarLenSparseFeat(namedtuple('VarLenSparseFeat', ('sparsefeat', 'maxlen', 'combiner', 'length_name'))):
    """A multi-valued categorical feature: ``maxlen`` id columns pooled by ``combiner``
    (reference inputs.py:41-77).  Every SparseFeat attribute is forwarded."""
    __slots__ = ()

    def __new__(cls, sparsefeat, maxlen, combiner="mean", length_name=None):
        return super().__new__(cls, sparsefeat, maxlen, combiner, length_name)

    name = property(lambda self: self.sparsefeat.name)
    vocabulary_size = property(lambda self: self.sparsefeat.vocabulary_size)
    embedding_dim = property(lambda self: self.sparsefeat.embedding_dim)
    use_hash = property(lambda self: self.sparsefeat.use_hash)
    dtype = property(lambda self: self.sparsefeat.dtype)
    embedding_name = property(lambda self: self.sparsefeat.embedding_name)
    group_name = property(lambda self: self.sparsefeat.group_name)

    def __hash__(self):
        return hash(self.name)


class DenseFeat(namedtuple('DenseFeat', ('name', 'dimension', 'dtype'))):
    """``dimension`` float columns of X used as they are (reference inputs.py:80-87)."""
    __slots__ = ()

    def __new__(cls, name, dimension=1, dtype="float32"):
        return super().__new__(cls, name, dimension, dtype)

    def __hash__(self):
        return hash(self.name)


def split_columns(feature_columns):
    """(SparseFeat list, VarLenSparseFeat list, DenseFeat list), each in declaration order."""
    cols = list(feature_columns) if feature_columns else []
    return ([c for c in cols if isinstance(c, SparseFeat)],
            [c for c in cols if isinstance(c, VarLenSparseFeat)],
            [c for c in cols if isinstance(c, DenseFeat)])


def build_input_features(feature_columns):
    """name -> (first column, one-past-last column) of the model input matrix X.
    Same layout rule as the reference (inputs.py:99-123): first occurrence of a name wins, a
    VarLenSparseFeat takes ``maxlen`` columns and its ``length_name`` (if new) one more."""
    layout = OrderedDict()
    cursor = 0
    for col in feature_columns:
        if not isinstance(col, (SparseFeat, DenseFeat, VarLenSparseFeat)):
            raise TypeError("Invalid feature column type,got", type(col))
        if col.name in layout:
            continue
        if isinstance(col, SparseFeat):
            width = 1
        elif isinstance(col, DenseFeat):
            width = col.dimension
        else:
            width = col.maxlen
        layout[col.name] = (cursor, cursor + width)
        cursor += width
        if isinstance(col, VarLenSparseFeat) and col.length_name is not None and col.length_name not in layout:
            layout[col.length_name] = (cursor, cursor + 1)
            cursor += 1
    return layout


def get_feature_names(feature_columns):
    return list(build_input_features(feature_columns).keys())


def create_embedding_matrix(feature_columns, init_std=0.0001, linear=False, sparse=False, device='cpu'):
    """``nn.ModuleDict{embedding_name: nn.Embedding}`` with N(0, init_std) weights -- the same
    parameters (and ``state_dict`` keys) as the reference (inputs.py:158-180).  The modules are only
    parameter holders here: lookups go through the fused kernel, never through ``nn.Embedding.forward``.
    """
    sparse_cols, varlen_cols, _ = split_columns(feature_columns)
    tables = nn.ModuleDict()
    for col in sparse_cols + varlen_cols:  # a later duplicate embedding_name replaces an earlier one
        tables[col.embedding_name] = nn.Embedding(col.vocabulary_size, 1 if linear else col.embedding_dim,
                                                  sparse=sparse)
    for emb in tables.values():
        nn.init.normal_(emb.weight, mean=0, std=init_std)
    return tables.to(device)


def combined_dnn_input(sparse_embedding_list, dense_value_list):
    """Flatten-and-concatenate (reference inputs.py:126-138).  The ★ models never call this -- the
    gather kernel writes this layout directly -- but out-of-scope models and user code may."""
    parts = []
    if len(sparse_embedding_list) > 0:
        parts.append(torch.flatten(torch.cat(list(sparse_embedding_list), dim=-1), start_dim=1))
    if len(dense_value_list) > 0:
        parts.append(torch.flatten(torch.cat(list(dense_value_list), dim=-1), start_dim=1))
    if not parts:
        raise NotImplementedError
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)


def get_dense_input(X, features, feature_columns):
    _, _, dense_cols = split_columns(feature_columns)
    return [X[:, features[fc.name][0]:features[fc.name][1]].float() for fc in dense_cols]


def maxlen_lookup(X, sparse_input_dict, maxlen_column):
    if maxlen_column is None or len(maxlen_column) == 0:
        raise ValueError('please add max length column for VarLenSparseFeat of DIN/DIEN input')
    lo, hi = sparse_input_dict[maxlen_column[0]]
    return X[:, lo:hi].long()


# --------------------------------------------------------------------------------------------------
# lookup helpers with the reference's signatures, routed through the HIP gather
# --------------------------------------------------------------------------------------------------
def embedding_lookup(X, sparse_embedding_dict, sparse_input_dict, sparse_feature_columns, return_feat_list=(),
                     mask_feat_list=(), to_list=False):
    """Per-feature ``[B, 1, D]`` embeddings grouped by ``group_name`` (reference inputs.py:183-210)."""
    from ._hip.ops import gather_columns
    wanted = [fc for fc in sparse_feature_columns
              if len(return_feat_list) == 0 or fc.name in return_feat_list]
    groups = defaultdict(list)
    if wanted:
        views = gather_columns(X, sparse_embedding_dict, sparse_input_dict, wanted, pooled=False)
        for fc, v in zip(wanted, views):
            groups[fc.group_name].append(v)
    if to_list:
        return list(chain.from_iterable(groups.values()))
    return groups


def varlen_embedding_lookup(X, embedding_dict, sequence_input_dict, varlen_sparse_feature_columns):
    """name -> un-pooled ``[B, maxlen, D]`` embeddings (reference inputs.py:213-227)."""
    from ._hip.ops import gather_columns
    cols = list(varlen_sparse_feature_columns)
    if not cols:
        return {}
    views = gather_columns(X, embedding_dict, sequence_input_dict, cols, pooled=False)
    return {fc.name: v for fc, v in zip(cols, views)}


def get_varlen_pooling_list(embedding_dict, features, feature_index, varlen_sparse_feature_columns, device):
    """Pooled ``[B, 1, D]`` per VarLen feature from the un-pooled dict (reference inputs.py:141-155)."""
    from .layers.sequence import SequencePoolingLayer
    pooled = []
    for feat in varlen_sparse_feature_columns:
        seq_emb = embedding_dict[feat.name]
        if feat.length_name is None:
            lo, hi = feature_index[feat.name]
            seq_mask = features[:, lo:hi].long() != 0
            emb = SequencePoolingLayer(mode=feat.combiner, supports_masking=True, device=device)([seq_emb, seq_mask])
        else:
            lo, hi = feature_index[feat.length_name]
            seq_length = features[:, lo:hi].long()
            emb = SequencePoolingLayer(mode=feat.combiner, supports_masking=False, device=device)(
                [seq_emb, seq_length])
        pooled.append(emb)
    return pooled

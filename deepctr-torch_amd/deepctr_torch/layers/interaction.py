"""Feature-interaction layers of the hot path (reference layers/interaction.py), each backed by a
hand-written gfx950 kernel in ``libdctr_hip.so``.  Constructor signatures, input/output shapes,
parameter names and error behaviour follow the reference so models and checkpoints drop in."""
import itertools

import torch
import torch.nn as nn

from .._hip import lib as _lib
from .._hip import ops as _ops
from .activation import Identity, activation_layer

__all__ = ["FM", "BiInteractionPooling", "AFMLayer", "InteractingLayer", "CrossNetMix", "CIN", "SENETLayer", "BilinearInteraction", "InnerProductLayer", "OutterProductLayer", "CrossNet"]


class FM(nn.Module):
    """Pairwise (order-2) interactions without linear term and bias:
    ``0.5 * sum_d((sum_f e)^2 - sum_f e^2)`` -- ``[B, F, D] -> [B, 1]`` (reference interaction.py:12-34).
    Kernel: ``dctr_fm_fwd`` / ``dctr_fm_bwd`` (csrc/fm.hip)."""

    def __init__(self):
        super(FM, self).__init__()

    def forward(self, inputs):
        return _ops.FMFunction.apply(inputs)


class BiInteractionPooling(nn.Module):
    """Bi-Interaction layer of Neural FM: the pairwise element-wise products of the fields compressed into one
    vector, ``0.5 * ((sum_f e)^2 - sum_f e^2)`` -- ``[B, F, D] -> [B, 1, D]`` (reference interaction.py:37-61).
    Kernel: ``dctr_bi_pooling_fwd`` / ``dctr_bi_pooling_bwd`` (csrc/fm.hip); NFM calls the fused form that also
    appends the dense features (``fused``)."""

    def __init__(self):
        super(BiInteractionPooling, self).__init__()

    def forward(self, inputs):
        if inputs.dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % inputs.dim())
        B, F, D = inputs.shape
        flat = inputs.reshape(B, F * D)
        return _ops.BiPoolFunction.apply(flat, F, D, F * D, 0)[:, :D].unsqueeze(1)

    @staticmethod
    def fused(gathered, F, D, dense_off, n_dense):
        """``gathered`` = the fused lookup's ``[B, ld]`` rows -> ``[B, r4(D + n_dense)]`` = ``[bi | dense]``."""
        return _ops.BiPoolFunction.apply(gathered, F, D, dense_off, n_dense)


class AFMLayer(nn.Module):
    """Attentional Factorization Machine pooling: ``softmax``-weighted sum of the pairwise element-wise products,
    projected to one logit -- list of F ``[B, 1, D]`` tensors (or one ``[B, F, D]`` tensor) ``-> [B, 1]`` (reference
    interaction.py:251-325; same constructor, same parameters ``attention_W [D, A]``, ``attention_b [A]``,
    ``projection_h [A, 1]``, ``projection_p [D, 1]``).  One kernel forward, one backward (``csrc/afm.hip``); with an
    active dropout on the attention output (the mask sits between two fused stages) or a shape outside the kernel
    (embedding_size > 64, attention_factor > 32, more than 64 fields) the same math runs as PyTorch-ROCm ops."""

    def __init__(self, in_features, attention_factor=4, l2_reg_w=0, dropout_rate=0, seed=1024, device='cpu'):
        super(AFMLayer, self).__init__()
        self.attention_factor = attention_factor
        self.l2_reg_w = l2_reg_w
        self.dropout_rate = dropout_rate
        self.seed = seed
        embedding_size = in_features
        self.attention_W = nn.Parameter(torch.Tensor(embedding_size, self.attention_factor))
        self.attention_b = nn.Parameter(torch.Tensor(self.attention_factor))
        self.projection_h = nn.Parameter(torch.Tensor(self.attention_factor, 1))
        self.projection_p = nn.Parameter(torch.Tensor(embedding_size, 1))
        for tensor in [self.attention_W, self.projection_h, self.projection_p]:
            nn.init.xavier_normal_(tensor, )
        for tensor in [self.attention_b]:
            nn.init.zeros_(tensor, )
        self.dropout = nn.Dropout(dropout_rate)
        self.to(device)

    @staticmethod
    def _kernel_fits(F, D, A):
        """csrc/afm.hip: D <= 64, A <= 32, F <= 64, and the BACKWARD's per-sample LDS image -- which holds every pair's
        attention row and gradient row -- within 150 KB (e.g. 55 fields of 16 with attention_factor 8)."""
        if F < 2:
            return True           # (the op itself rejects a single field, like the reference's empty torch.cat)
        if D > 64 or A > 32 or F > 64:
            return False
        ap = 4
        while ap < A:
            ap <<= 1
        P = F * (F - 1) // 2
        n = D * ap + 2 * ap + D + P + F * D + P + D + (D + P * ap + P * D)
        return 4 * n <= 150 * 1024

    def forward(self, inputs):
        E = torch.cat(list(inputs), dim=1) if isinstance(inputs, (list, tuple)) else inputs
        if E.dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % E.dim())
        F_, D_ = E.shape[1], E.shape[2]
        outside = not self._kernel_fits(F_, D_, self.attention_factor)
        if (self.dropout_rate and self.training) or outside:
            F_ = E.shape[1]
            idx = torch.triu_indices(F_, F_, 1, device=E.device)
            bi = E[:, idx[0]] * E[:, idx[1]]
            att = torch.relu(torch.tensordot(bi, self.attention_W, dims=([-1], [0])) + self.attention_b)
            score = torch.softmax(torch.tensordot(att, self.projection_h, dims=([-1], [0])), dim=1)
            out = self.dropout(torch.sum(score * bi, dim=1))
            return torch.tensordot(out, self.projection_p, dims=([-1], [0]))
        return _ops.AFMFunction.apply(E, self.attention_W, self.attention_b, self.projection_h, self.projection_p)


class InteractingLayer(nn.Module):
    """Multi-head self-attention over the fields (AutoInt): ``[B, F, D] -> [B, F, D]`` (reference
    interaction.py:328-394; same constructor, same ``W_Query / W_key / W_Value / W_Res [D, D]`` parameters).

    One kernel forward, one backward (``csrc/interact.hip``): a wave owns a sample, E / Q / K / V / the H score matrices
    stay in LDS, the backward recomputes the forward and sums the weight gradients in a fixed order.  Shapes outside
    the kernel (embedding_size > 32, more than 64 fields) run the same arithmetic as batched GEMMs on PyTorch-ROCm."""

    def __init__(self, embedding_size, head_num=2, use_res=True, scaling=False, seed=1024, device='cpu'):
        super(InteractingLayer, self).__init__()
        if head_num <= 0:
            raise ValueError('head_num must be a int > 0')
        if embedding_size % head_num != 0:
            raise ValueError('embedding_size is not an integer multiple of head_num!')
        self.att_embedding_size = embedding_size // head_num
        self.head_num = head_num
        self.use_res = use_res
        self.scaling = scaling
        self.seed = seed
        self.W_Query = nn.Parameter(torch.Tensor(embedding_size, embedding_size))
        self.W_key = nn.Parameter(torch.Tensor(embedding_size, embedding_size))
        self.W_Value = nn.Parameter(torch.Tensor(embedding_size, embedding_size))
        if self.use_res:
            self.W_Res = nn.Parameter(torch.Tensor(embedding_size, embedding_size))
        for tensor in self.parameters():
            nn.init.normal_(tensor, mean=0.0, std=0.05)
        self.to(device)

    def forward(self, inputs):
        if len(inputs.shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(inputs.shape)))
        B, F_, D = inputs.shape
        H, A = self.head_num, self.att_embedding_size
        if inputs.is_cuda and _ops.interacting_supported(F_, D, H):
            return _ops.InteractFunction.apply(inputs, self.W_Query, self.W_key, self.W_Value,
                                               self.W_Res if self.use_res else None, H, self.scaling)
        # [B, F, D] -> [B, H, F, A]: head n = columns n*A .. (n+1)*A (torch.split(..., dim=2) of the reference)
        q = torch.matmul(inputs, self.W_Query).view(B, F_, H, A).transpose(1, 2)
        k = torch.matmul(inputs, self.W_key).view(B, F_, H, A).transpose(1, 2)
        v = torch.matmul(inputs, self.W_Value).view(B, F_, H, A).transpose(1, 2)
        inner = torch.matmul(q, k.transpose(2, 3))                       # [B, H, F, F]
        if self.scaling:
            inner = inner / self.att_embedding_size ** 0.5
        self.normalized_att_scores = torch.softmax(inner, dim=-1)
        result = torch.matmul(self.normalized_att_scores, v)             # [B, H, F, A]
        result = result.transpose(1, 2).reshape(B, F_, D)                # heads side by side (torch.cat(..., dim=-1))
        if self.use_res:
            result = result + torch.matmul(inputs, self.W_Res)
        return torch.relu(result)


class CrossNetMix(nn.Module):
    """Cross network of DCN-Mix: a mixture of low-rank experts per layer, ``[B, W] -> [B, W]`` (reference
    interaction.py:456-534; same constructor, same ``U_list / V_list / C_list [L, E, ...]``, ``gating.<e>.weight``,
    ``bias [L, W, 1]`` parameters).  All layers in one fp32-MFMA launch each way (csrc/mlp.hip,
    ``dctr_crossnet_mix_*``: three dense layers per cross layer on a 16-sample tile in LDS; up to 512 inputs, 4 cross
    layers, 8 experts); beyond that the batched-einsum formulation on PyTorch-ROCm."""

    def __init__(self, in_features, low_rank=32, num_experts=4, layer_num=2, device='cpu'):
        super(CrossNetMix, self).__init__()
        self.layer_num = layer_num
        self.num_experts = num_experts
        self.U_list = nn.Parameter(torch.Tensor(self.layer_num, num_experts, in_features, low_rank))
        self.V_list = nn.Parameter(torch.Tensor(self.layer_num, num_experts, in_features, low_rank))
        self.C_list = nn.Parameter(torch.Tensor(self.layer_num, num_experts, low_rank, low_rank))
        self.gating = nn.ModuleList([nn.Linear(in_features, 1, bias=False) for i in range(self.num_experts)])
        self.bias = nn.Parameter(torch.Tensor(self.layer_num, in_features, 1))
        for para in [self.U_list, self.V_list, self.C_list]:
            for i in range(self.layer_num):
                nn.init.xavier_normal_(para[i])
        for i in range(len(self.bias)):
            nn.init.zeros_(self.bias[i])
        self.to(device)

    def forward(self, inputs):
        gate_w = torch.cat([g.weight for g in self.gating], dim=0)       # [E, W]
        if inputs.dim() == 2 and inputs.dtype == torch.float32 and self.layer_num > 0 and \
                _lib.lib().dctr_crossnet_mix_supported(int(inputs.shape[1]), int(self.layer_num),
                                                       int(self.num_experts), int(self.U_list.shape[3])):
            return _ops.CrossNetMixFunction.apply(inputs, self.U_list, self.V_list, self.C_list, gate_w, self.bias)
        x_0 = inputs                                                     # [B, W]: beyond the kernels' LDS tiles
        x_l = x_0
        for i in range(self.layer_num):
            score = torch.softmax(torch.matmul(x_l, gate_w.t()), dim=1)  # [B, E]   G(x_l)
            v_x = torch.tanh(torch.einsum("bw,ewr->ber", x_l, self.V_list[i]))          # project to R^r
            v_x = torch.tanh(torch.einsum("ers,bes->ber", self.C_list[i], v_x))
            uv_x = torch.einsum("ewr,ber->bew", self.U_list[i], v_x)                    # back to R^W
            dot_ = x_0.unsqueeze(1) * (uv_x + self.bias[i].squeeze(1))                  # Hadamard product, [B, E, W]
            x_l = torch.einsum("bew,be->bw", dot_, score) + x_l                         # mixture of the experts
        return x_l


class CIN(nn.Module):
    """Compressed Interaction Network of xDeepFM: ``[B, F, D] -> [B, featuremap_num]`` (reference
    interaction.py:159-248; same constructor, same ``conv1ds.<k>.weight [O, h*F, 1]`` / ``bias`` parameters).

    Each layer is ONE fp32-MFMA kernel (``csrc/cin.hip``) that never materialises the reference's
    ``[B, h*F, D]`` outer product (436 MB at the Criteo shape); relu (the default) is fused, any other activation
    module is applied to the kernel's linear output."""

    def __init__(self, field_size, layer_size=(128, 128), activation='relu', split_half=True, l2_reg=1e-5, seed=1024,
                 device='cpu'):
        super(CIN, self).__init__()
        if len(layer_size) == 0:
            raise ValueError("layer_size must be a list(tuple) of length greater than 1")
        self.layer_size = layer_size
        self.field_nums = [field_size]
        self.split_half = split_half
        self.activation = activation_layer(activation)
        self.l2_reg = l2_reg
        self.seed = seed
        self.conv1ds = nn.ModuleList()
        for i, size in enumerate(self.layer_size):
            self.conv1ds.append(nn.Conv1d(self.field_nums[-1] * self.field_nums[0], size, 1))
            if self.split_half:
                if i != len(self.layer_size) - 1 and size % 2 > 0:
                    raise ValueError("layer_size must be even number except for the last layer when split_half=True")
                self.field_nums.append(size // 2)
            else:
                self.field_nums.append(size)
        self.to(device)

    def _stack_ok(self, n_fields):
        """The common case on the kernels' own terms (csrc/cin.hip: at most 32 fields, relu or no activation)."""
        plain = self.activation is None or type(self.activation) in (nn.ReLU, Identity, nn.Identity)
        return n_fields <= 32 and plain and not (self.activation is not None and (
            self.activation._forward_hooks or self.activation._forward_pre_hooks))

    def _stack(self, x, F, D, w_head):
        """The whole stack as one autograd node (_hip/ops.py CINStackFunction): ``x`` the ``[B, F, D]`` field matrix or a
        ``[B, >= F*D]`` row matrix that starts with it; ``w_head``: None, or the ``[1, featuremap_num]`` weight of the
        bias-free projection a model puts on the output (xDeepFM's cin_linear) -- the result is then ``[B, 1]``."""
        wb = []
        for conv in self.conv1ds:
            wb += [conv.weight, conv.bias]
        return _ops.CINStackFunction.apply(x, F, D, isinstance(self.activation, nn.ReLU), bool(self.split_half), w_head,
                                           *wb)

    def forward(self, inputs):
        if len(inputs.shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(inputs.shape)))
        x0 = inputs
        hidden = x0
        fused_relu = isinstance(self.activation, nn.ReLU)
        if self._stack_ok(x0.shape[1]):
            return self._stack(x0, x0.shape[1], x0.shape[2], None)
        final_result = []
        for i, size in enumerate(self.layer_size):
            conv = self.conv1ds[i]
            M = x0.shape[1]
            if M <= 32:
                curr_out = _ops.CINLayerFunction.apply(hidden, x0, conv.weight.squeeze(-1), conv.bias, fused_relu)
            else:
                # More fields than one MFMA tile of csrc/cin.hip (M <= 32), e.g. Criteo with its 13 dense columns
                # bucketised into fields (39): the sum over the field index m of x_0 splits into groups of <= 32
                # fields -- one kernel call per group on the group's slice of x_0 (a view) and of the filter, the
                # pre-activations added, then bias / relu.  Everything stays on the MFMA kernels.
                W3 = conv.weight.squeeze(-1).view(size, hidden.shape[1], M)
                curr_out = None
                for lo in range(0, M, 32):
                    hi = min(M, lo + 32)
                    part = _ops.CINLayerFunction.apply(hidden, x0[:, lo:hi], W3[:, :, lo:hi].reshape(size, -1),
                                                       conv.bias if lo == 0 else None, False)
                    curr_out = part if curr_out is None else curr_out + part
                if fused_relu:
                    curr_out = torch.relu(curr_out)
            if not fused_relu and self.activation is not None:
                curr_out = self.activation(curr_out)
            if self.split_half:
                if i != len(self.layer_size) - 1:
                    hidden, direct_connect = torch.split(curr_out, 2 * [size // 2], 1)
                else:
                    direct_connect, hidden = curr_out, None
            else:
                direct_connect, hidden = curr_out, curr_out
            final_result.append(direct_connect)
        return torch.sum(torch.cat(final_result, dim=1), -1)


class SENETLayer(nn.Module):
    """Squeeze-and-excitation over fields: ``[B, F, D] -> [B, F, D]`` (reference interaction.py:64-101; parameters
    ``excitation.0.weight [F//r, F]``, ``excitation.2.weight [F, F//r]``).  One kernel each way (csrc/pairwise.hip)."""

    def __init__(self, filed_size, reduction_ratio=3, seed=1024, device='cpu'):
        super(SENETLayer, self).__init__()
        self.seed = seed
        self.filed_size = filed_size
        self.reduction_size = max(1, filed_size // reduction_ratio)
        self.excitation = nn.Sequential(
            nn.Linear(self.filed_size, self.reduction_size, bias=False), nn.ReLU(),
            nn.Linear(self.reduction_size, self.filed_size, bias=False), nn.ReLU())
        self.to(device)

    def forward(self, inputs):
        if len(inputs.shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(inputs.shape)))
        F_, D_ = inputs.shape[1], inputs.shape[2]
        if 16 * F_ * D_ + 24 * F_ + 8 * self.reduction_size > 16384:
            # the backward kernel stages 2 x 8 samples x F x D floats in 64 KB of LDS (csrc/pairwise.hip,
            # dctr_senet_bwd): beyond F*D ~ 1000 (e.g. 39 fields of 32) the reference's formulation on PyTorch-ROCm
            A = self.excitation(torch.mean(inputs, dim=-1))
            return torch.mul(inputs, torch.unsqueeze(A, dim=2))
        return _ops.SENETFunction.apply(inputs, self.excitation[0].weight, self.excitation[2].weight)


class BilinearInteraction(nn.Module):
    """``p_k = (v_i W) * v_j`` for every field pair: ``[B, F, D] -> [B, F(F-1)/2, D]`` (reference
    interaction.py:104-156; parameters ``bilinear.weight`` / ``bilinear.<i>.weight`` / ``bilinear.<k>.weight``).
    All pairs run in one fp32-MFMA kernel instead of 325 ``nn.Linear`` calls and a 325-way ``cat``."""

    def __init__(self, filed_size, embedding_size, bilinear_type="interaction", seed=1024, device='cpu'):
        super(BilinearInteraction, self).__init__()
        self.bilinear_type = bilinear_type
        self.seed = seed
        self.filed_size = filed_size
        self.bilinear = nn.ModuleList()
        if self.bilinear_type == "all":
            self.bilinear = nn.Linear(embedding_size, embedding_size, bias=False)
        elif self.bilinear_type == "each":
            for _ in range(filed_size):
                self.bilinear.append(nn.Linear(embedding_size, embedding_size, bias=False))
        elif self.bilinear_type == "interaction":
            for _, _ in itertools.combinations(range(filed_size), 2):
                self.bilinear.append(nn.Linear(embedding_size, embedding_size, bias=False))
        else:
            raise NotImplementedError
        self._meta = None
        self.to(device)

    def _weights(self):
        if self.bilinear_type == "all":
            return [self.bilinear.weight]
        return [lin.weight for lin in self.bilinear]

    def meta(self, n_fields):
        if self._meta is None or self._meta[0] != n_fields:
            self._meta = (n_fields, _ops.BilinearMeta(n_fields, self.bilinear_type))
        return self._meta[1]

    def stacked_weights(self):
        """(parameters, their shared ``[n_w, D, D]`` slab) once the kernels have re-seated the per-pair ``nn.Linear``
        weights as slices of one slab (first forward), else None.  BaseModel steps such a group with ONE optimizer
        launch instead of handing hundreds of tiny tensors to ``torch.optim``'s foreach kernels."""
        if self._meta is None or self.bilinear_type == "all":
            return None
        slab = self._meta[1]._slab
        ws = self._weights()
        if slab is None or slab.shape[0] != len(ws) or ws[0].data_ptr() != slab.data_ptr() or \
                ws[-1].data_ptr() != slab[-1].data_ptr():
            return None
        return ws, slab

    @staticmethod
    def _kernel_fits(F, D):
        """The backward-data kernel keeps 4 tiles of 16 samples x (F*D padded) floats in LDS (csrc/pairwise.hip,
        dctr_bilinear_bwd: <= 158 KB) and the MFMA tiling needs D <= 16: F*D up to ~600, i.e. 37 fields of 16."""
        rs = F * D
        rs += (16 - (rs & 31)) & 31
        return D <= 16 and 4 * 16 * rs * 4 + 4 * 16 * 17 * 4 <= 158 * 1024

    def _pairs_torch(self, X):
        """``[B, F, D] -> [B, P, D]`` as batched GEMMs on PyTorch-ROCm, for shapes outside the MFMA kernels
        (``_kernel_fits``)."""
        F = X.shape[1]
        idx = torch.triu_indices(F, F, 1, device=X.device)
        left, right = X[:, idx[0]], X[:, idx[1]]
        if self.bilinear_type == "all":
            return torch.matmul(left, self.bilinear.weight.t()) * right
        W = torch.stack(self._weights())                       # [F | P, D_out, D_in]
        Wp = W[idx[0]] if self.bilinear_type == "each" else W
        return torch.einsum("bpd,ped->bpe", left, Wp) * right

    def forward(self, inputs):
        if len(inputs.shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(inputs.shape)))
        B, F, D = inputs.shape
        if not self._kernel_fits(F, D):
            return self._pairs_torch(inputs)
        out = _ops.BilinearFunction.apply(self.meta(F), inputs, None, None, *self._weights())
        return out.reshape(B, F * (F - 1) // 2, D)

    def fused_pair(self, raw, senet, dense=None, lazy=False):
        """FiBiNET's ``cat(Bilinear(senet), Bilinear(raw))`` flattened, followed by the dense features: the DNN input
        of fibinet.py:82-87 produced by one launch that loads every weight tile once for both passes.  ``lazy``: hand
        the tower a ``PendingPairs`` instead (pairs + first tower layer as one autograd node, csrc/bilinear_wide.hip)."""
        if lazy and raw.is_cuda and self._kernel_fits(raw.shape[1], raw.shape[2]):
            from .._hip import mlp as _mlp
            return _mlp.PendingPairs(self.meta(raw.shape[1]), raw, senet, dense, self._weights())
        if not self._kernel_fits(raw.shape[1], raw.shape[2]):
            parts = [self._pairs_torch(senet).flatten(1), self._pairs_torch(raw).flatten(1)]
            return torch.cat(parts + ([dense] if dense is not None else []), dim=1)
        return _ops.BilinearFunction.apply(self.meta(raw.shape[1]), raw, senet, dense, *self._weights())


def pairwise_products(E, reduce_sum):
    """``e_i * e_j`` for i < j: ``[B, F, D] -> [B, P, 1 | D]``.  The kernels (csrc/pairwise.hip) keep one sample's fields
    and -- in the backward -- its P gradient rows in 60 KB of LDS; beyond that (e.g. 45 fields of 16 without the sum)
    the same products as PyTorch-ROCm ops."""
    F_, D_ = E.shape[1], E.shape[2]
    P = F_ * (F_ - 1) // 2
    if F_ < 2 or (F_ * D_ + P * (1 if reduce_sum else D_) + 2) * 4 <= 60 * 1024:
        return _ops.InnerProductFunction.apply(E, reduce_sum)
    idx = torch.triu_indices(F_, F_, 1, device=E.device)
    prod = E[:, idx[0]] * E[:, idx[1]]
    return prod.sum(dim=2, keepdim=True) if reduce_sum else prod


class InnerProductLayer(nn.Module):
    """Pairwise inner (or element-wise) products of field embeddings (reference interaction.py:537-577):
    list of ``[B, 1, D]`` (or one ``[B, F, D]`` tensor) -> ``[B, F(F-1)/2, 1]`` (``[.., D]`` without reduce_sum)."""

    def __init__(self, reduce_sum=True, device='cpu'):
        super(InnerProductLayer, self).__init__()
        self.reduce_sum = reduce_sum
        self.to(device)

    def forward(self, inputs):
        E = inputs if torch.is_tensor(inputs) else torch.cat(list(inputs), dim=1)
        return pairwise_products(E, self.reduce_sum)


class OutterProductLayer(nn.Module):
    """Outer-product layer of PNN: per pair (i < j) the scalar ``p^T K q`` for the kernel types ``mat`` (K = ``kernel[:, k,
    :]``, a full D x D form), ``vec`` (diagonal) and ``num`` (a scalar) -- list of F ``[B, 1, D]`` tensors (or one
    ``[B, F, D]`` tensor) ``-> [B, F(F-1)/2]`` (reference interaction.py:580-672; same constructor, same ``kernel``
    parameter).  'mat' is the bilinear form ``sum_e' (x_i W_k^T)[e'] x_j[e']`` with ``W_k = kernel[:, k, :]``: it runs on
    the MFMA bilinear kernels of csrc/pairwise.hip followed by a row sum; 'vec' / 'num' scale the element-wise products
    of csrc/pairwise.hip's inner-product kernel.  (embedding_size > 16 with 'mat': batched GEMMs on PyTorch-ROCm.)"""

    def __init__(self, field_size, embedding_size, kernel_type='mat', seed=1024, device='cpu'):
        super(OutterProductLayer, self).__init__()
        self.kernel_type = kernel_type
        num_inputs = field_size
        num_pairs = int(num_inputs * (num_inputs - 1) / 2)
        embed_size = embedding_size
        if self.kernel_type == 'mat':
            self.kernel = nn.Parameter(torch.Tensor(embed_size, num_pairs, embed_size))
        elif self.kernel_type == 'vec':
            self.kernel = nn.Parameter(torch.Tensor(num_pairs, embed_size))
        elif self.kernel_type == 'num':
            self.kernel = nn.Parameter(torch.Tensor(num_pairs, 1))
        nn.init.xavier_uniform_(self.kernel)
        self._meta = None
        self.to(device)

    def forward(self, inputs):
        E = inputs if torch.is_tensor(inputs) else torch.cat(list(inputs), dim=1)
        B, F_, D = E.shape
        P = F_ * (F_ - 1) // 2
        if self.kernel_type == 'mat':
            if D <= 16:
                if self._meta is None or self._meta[0] != F_:
                    self._meta = (F_, _ops.BilinearMeta(F_, "interaction"))
                prod = _ops.BilinearStackedFunction.apply(self._meta[1], E, self.kernel.permute(1, 0, 2))
                return prod.view(B, P, D).sum(dim=-1)
            idx = torch.triu_indices(F_, F_, 1, device=E.device)
            pk = torch.einsum("bke,fke->bkf", E[:, idx[0]], self.kernel)
            return torch.sum(pk * E[:, idx[1]], dim=-1)
        prod = pairwise_products(E, False)                               # [B, P, D]: p (.) q
        return torch.sum(prod * self.kernel.unsqueeze(0), dim=-1)


class CrossNet(nn.Module):
    """Cross network of DCN / DCN-M: ``[B, W] -> [B, W]`` (reference interaction.py:397-453; parameters
    ``kernels [L, W, 1 | W]``, ``bias [L, W, 1]``).  The vector form runs all layers in one wave-per-sample kernel
    (csrc/cross.hip); the matrix form runs all layers in one fp32-MFMA launch that keeps x_0 and x_l of a 16-sample
    tile in LDS (csrc/mlp.hip, ``dctr_crossnet_mat_*``; up to 512 inputs, wider ones go to hipBLASLt)."""

    def __init__(self, in_features, layer_num=2, parameterization='vector', seed=1024, device='cpu'):
        super(CrossNet, self).__init__()
        self.layer_num = layer_num
        self.parameterization = parameterization
        if self.parameterization == 'vector':
            self.kernels = nn.Parameter(torch.Tensor(self.layer_num, in_features, 1))
        elif self.parameterization == 'matrix':
            self.kernels = nn.Parameter(torch.Tensor(self.layer_num, in_features, in_features))
        else:
            raise ValueError("parameterization should be 'vector' or 'matrix'")
        self.bias = nn.Parameter(torch.Tensor(self.layer_num, in_features, 1))
        for i in range(self.kernels.shape[0]):
            nn.init.xavier_normal_(self.kernels[i])
        for i in range(self.bias.shape[0]):
            nn.init.zeros_(self.bias[i])
        self.to(device)

    def forward(self, inputs):
        if self.layer_num == 0:
            return inputs
        W_ = inputs.shape[1]
        # csrc/cross.hip: a wave holds a sample of up to 2048 floats; the backward keeps 4 x layers x W floats in LDS
        fits = W_ <= 2048 and (4 * self.layer_num * W_ + 4 * self.layer_num) * 4 <= 150 * 1024
        if self.parameterization == 'vector' and fits:
            return _ops.CrossNetVecFunction.apply(inputs, self.kernels, self.bias)
        if self.parameterization == 'vector':          # outside the kernels' envelope: PyTorch-ROCm ops
            x_0 = x_l = inputs
            for i in range(self.layer_num):            # x0 * (x_l . w) + b + x_l
                x_l = x_0 * torch.matmul(x_l, self.kernels[i]) + self.bias[i].squeeze(1) + x_l
            return x_l
        if self.parameterization == 'matrix' and inputs.dim() == 2 and inputs.dtype == torch.float32 and \
                _lib.lib().dctr_crossnet_mat_supported(int(W_), int(self.layer_num)):
            return _ops.CrossNetMatFunction.apply(inputs, self.kernels, self.bias)
        x_0 = inputs                      # wider than the kernels hold in LDS (> 512 inputs): PyTorch-ROCm GEMMs
        x_l = x_0
        for i in range(self.layer_num):   # x0 * (W x_l + b) + x_l
            x_l = x_0 * (torch.addmm(self.bias[i].t(), x_l, self.kernels[i].t())) + x_l
        return x_l

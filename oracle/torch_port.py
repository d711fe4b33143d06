"""CPU baseline: the reference's DeepFM training step restated with plain torch-CPU modules.

TEST / BENCH INFRASTRUCTURE ONLY (``bench.py``'s ``cpu_baseline`` leg; ``kind = "port"``).  The reference
cannot travel to the GPU box, and its arithmetic is nothing but ATen calls, so this file issues the same ATen
calls in the same order: one ``nn.Embedding`` per table with DENSE ``[V, D]`` gradients (inputs.py:168,
``sparse=False``), per-field lookups in Python loops (basemodel.py:65-67,368-370), ``torch.cat`` + FM
(deepfm.py:74-75, interaction.py:26-34), ``DNN`` (core.py:120-134), ``binary_cross_entropy(reduction='sum')``
(basemodel.py:254) and a dense ``torch.optim`` step over every parameter (basemodel.py:262).  It is pinned to
the golden fixtures in tests/test_oracle_golden.py::test_torch_port_matches_reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class DeepFMPort(nn.Module):
    def __init__(self, n_sparse, vocab, dim, n_dense, hidden=(256, 128), init_std=1e-4, seed=1024):
        super().__init__()
        torch.manual_seed(seed)
        vocabs = vocab if isinstance(vocab, (list, tuple)) else [vocab] * n_sparse
        self.n_sparse, self.n_dense = n_sparse, n_dense
        self.emb = nn.ModuleList(nn.Embedding(v, dim) for v in vocabs)
        self.lin = nn.ModuleList(nn.Embedding(v, 1) for v in vocabs)
        for e in list(self.emb) + list(self.lin):
            nn.init.normal_(e.weight, mean=0, std=init_std)
        self.lin_w = nn.Parameter(torch.empty(n_dense, 1).normal_(0, init_std))
        widths = [n_sparse * dim + n_dense] + list(hidden)
        self.linears = nn.ModuleList(nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))
        for l in self.linears:
            nn.init.normal_(l.weight, mean=0, std=init_std)
        self.dnn_linear = nn.Linear(widths[-1], 1, bias=False)
        self.bias = nn.Parameter(torch.zeros(1))

    def logit(self, X):
        ids = [X[:, f:f + 1].long() for f in range(self.n_sparse)]
        dense = X[:, self.n_sparse:self.n_sparse + self.n_dense]
        embs = [self.emb[f](ids[f]) for f in range(self.n_sparse)]          # [B, 1, D] each
        lins = [self.lin[f](ids[f]) for f in range(self.n_sparse)]          # [B, 1, 1] each
        logit = torch.zeros([X.shape[0], 1]) + torch.sum(torch.cat(lins, dim=-1), dim=-1) + dense.matmul(self.lin_w)
        fm_in = torch.cat(embs, dim=1)
        sq_of_sum = torch.pow(torch.sum(fm_in, dim=1, keepdim=True), 2)
        sum_of_sq = torch.sum(fm_in * fm_in, dim=1, keepdim=True)
        logit = logit + 0.5 * torch.sum(sq_of_sum - sum_of_sq, dim=2, keepdim=False)
        h = torch.cat([torch.flatten(torch.cat(embs, dim=-1), start_dim=1), dense], dim=-1)
        for l in self.linears:
            h = torch.relu(l(h))
        return logit + self.dnn_linear(h)

    def forward(self, X):
        return torch.sigmoid(self.logit(X) + self.bias)

    def load_reference_state(self, params, names):
        """Copy a reference ``state_dict`` (numpy arrays keyed by the reference's keys)."""
        with torch.no_grad():
            for f, n in enumerate(names):
                self.emb[f].weight.copy_(torch.as_tensor(params["embedding_dict.%s.weight" % n]))
                self.lin[f].weight.copy_(torch.as_tensor(params["linear_model.embedding_dict.%s.weight" % n]))
            self.lin_w.copy_(torch.as_tensor(params["linear_model.weight"]))
            for i, l in enumerate(self.linears):
                l.weight.copy_(torch.as_tensor(params["dnn.linears.%d.weight" % i]))
                l.bias.copy_(torch.as_tensor(params["dnn.linears.%d.bias" % i]))
            self.dnn_linear.weight.copy_(torch.as_tensor(params["dnn_linear.weight"]))
            self.bias.copy_(torch.as_tensor(params["out.bias"]))


def make_optimizer(model, name):
    if name == "sgd":
        return torch.optim.SGD(model.parameters(), lr=0.01)
    if name == "adagrad":
        return torch.optim.Adagrad(model.parameters())
    if name == "adam":
        return torch.optim.Adam(model.parameters())
    raise ValueError(name)


def reg_loss(model, l2_embedding, l2_linear):
    """basemodel.py:412-428 as DeepFM registers it (basemodel.py:124-127, deepfm.py:55-58): l2_reg_embedding on the deep
    tables, l2_reg_linear on everything of the linear model; the same ATen calls (square, scalar mul, sum, in-place add)."""
    total = torch.zeros((1,))
    if l2_embedding > 0:
        for e in model.emb:
            total += torch.sum(l2_embedding * torch.square(e.weight))
    if l2_linear > 0:
        for e in model.lin:
            total += torch.sum(l2_linear * torch.square(e.weight))
        total += torch.sum(l2_linear * torch.square(model.lin_w))
    return total


def train_step(model, optim, X, y, l2_embedding=0.0, l2_linear=0.0):
    """basemodel.py:242-262 (the reference adds get_regularization_loss() even when it is identically zero)."""
    y_pred = model(X).squeeze()
    optim.zero_grad()
    loss = F.binary_cross_entropy(y_pred, y.squeeze(), reduction='sum')
    total = loss + reg_loss(model, l2_embedding, l2_linear)
    total.backward()
    optim.step()
    return loss

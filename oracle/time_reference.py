"""Time the UNMODIFIED reference (shenweichen/DeepCTR-Torch, torch-CPU fp32) on the bench workload, in the build
container (the only place /root/reference exists): DeepFM, 26 sparse x 1M vocab + 13 dense, emb_dim 16, batch 4096,
the inner train step exactly as basemodel.py:242-262 (forward, BCE(sum) + regularisation, backward, optimizer step).
SURVEY.md 8(d): (i) the reference's defaults (l2 = 1e-5, adam), (ii) like-for-like with bench.py (l2 = 0, adagrad),
(iii) forward only.  2 warm-up + 5 timed steps each.  Also times oracle/torch_port.py (what bench.py's cpu_baseline leg
runs on the GPU box) on the same cores, so the two baselines can be related.
    python oracle/time_reference.py > profiles/r02_reference_cpu_timing.json"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

F, V, D, ND, B = 26, 1_000_000, 16, 13, 4096


def main():
    ref = mg.import_reference()
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    cols = [SparseFeat("C%d" % (i + 1), V, D) for i in range(F)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(ND)]
    gen = torch.Generator().manual_seed(0)
    X = torch.cat([torch.randint(0, V, (B, F), generator=gen).float(), torch.rand(B, ND, generator=gen)], 1)
    y = torch.randint(0, 2, (B,), generator=gen).float()
    out = {"host": {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__},
           "workload": "DeepFM Criteo-shaped: 26 x 1M x 16, 13 dense, batch 4096, dnn (256,128)", "runs": {}}

    def step(model, optim):
        y_pred = model(X).squeeze()
        optim.zero_grad()
        loss = torch.nn.functional.binary_cross_entropy(y_pred, y, reduction="sum")
        total = loss + model.get_regularization_loss() + model.aux_loss
        total.backward()
        optim.step()

    for tag, l2, opt in (("reference_defaults_l2_1e-5_adam", 1e-5, "adam"), ("like_for_like_l2_0_adagrad", 0.0, "adagrad")):
        m = DeepFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=l2, l2_reg_embedding=l2, dnn_dropout=0,
                   seed=1024, device="cpu")
        m.compile(opt, "binary_crossentropy", metrics=[])
        m.train()
        for _ in range(2):
            step(m, m.optim)
        t0 = time.perf_counter()
        for _ in range(5):
            step(m, m.optim)
        dt = (time.perf_counter() - t0) / 5
        out["runs"][tag] = {"ms_per_step": dt * 1e3, "samples_per_s": B / dt}
        if opt == "adagrad":
            m.eval()
            with torch.no_grad():
                for _ in range(2):
                    m(X)
                t0 = time.perf_counter()
                for _ in range(5):
                    m(X)
                dtf = (time.perf_counter() - t0) / 5
            out["runs"]["forward_only"] = {"ms_per_step": dtf * 1e3, "samples_per_s": B / dtf}
        del m
    from torch_port import DeepFMPort, make_optimizer, train_step
    torch.manual_seed(0)
    port = DeepFMPort(F, V, D, ND, hidden=(256, 128))
    popt = make_optimizer(port, "adagrad")
    for _ in range(2):
        train_step(port, popt, X, y)
    t0 = time.perf_counter()
    for _ in range(5):
        train_step(port, popt, X, y)
    dt = (time.perf_counter() - t0) / 5
    out["runs"]["torch_port_l2_0_adagrad (bench.py cpu_baseline kind=port)"] = {"ms_per_step": dt * 1e3,
                                                                                "samples_per_s": B / dt}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""Time the UNMODIFIED reference (shenweichen/DeepCTR-Torch, torch-CPU fp32) on the bench workload, in the build
container (the only place /root/reference exists): DeepFM, 26 sparse x 1M vocab + 13 dense, emb_dim 16, batch 4096,
the inner train step exactly as basemodel.py:242-262 (forward, BCE(sum) + regularisation, backward, optimizer step).
SURVEY.md 8(d): (i) the reference's defaults (l2 = 1e-5, adam), (ii) like-for-like with bench.py (l2 = 0, adagrad),
(iii) forward only -- and the same three variants of oracle/torch_port.py (what bench.py's cpu_baseline leg runs on the
GPU box), so that the two baselines can be related.  3 warm-up + 5 timed steps each.

EVERY variant runs in its own process: a step allocates a fresh 1.8 GB dense gradient (zero_grad(set_to_none=True)), and
what the allocator and the page cache hold from an earlier model moves a step time by 2-4x (round 2 timed the port
last in one process and read 2.8 s / step; alone it takes 0.75 s against the reference's 0.70).

    python oracle/time_reference.py > profiles/r03_reference_cpu_timing.json"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
F, V, D, ND, B = 26, 1_000_000, 16, 13, 4096
VARIANTS = (("like_for_like_l2_0_adagrad", 0.0, "adagrad"), ("reference_defaults_l2_1e-5_adam", 1e-5, "adam"))


def data():
    import torch
    gen = torch.Generator().manual_seed(0)
    X = torch.cat([torch.randint(0, V, (B, F), generator=gen).float(), torch.rand(B, ND, generator=gen)], 1)
    y = torch.randint(0, 2, (B,), generator=gen).float()
    return X, y


def timed(fn, n=5, warm=3):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n
    return {"ms_per_step": dt * 1e3, "samples_per_s": B / dt}


def one(which, tag):
    import torch
    X, y = data()
    l2, opt = {t: (l, o) for t, l, o in VARIANTS}.get(tag, (0.0, "adagrad"))
    if which == "port":
        sys.path.insert(0, HERE)
        from torch_port import DeepFMPort, make_optimizer, train_step
        torch.manual_seed(0)
        port = DeepFMPort(F, V, D, ND, hidden=(256, 128))
        if tag == "forward_only":
            with torch.no_grad():
                return timed(lambda: port(X))
        popt = make_optimizer(port, opt)
        return timed(lambda: train_step(port, popt, X, y, l2, l2))
    sys.path.insert(0, HERE)
    import make_golden as mg
    mg.import_reference()
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    cols = [SparseFeat("C%d" % (i + 1), V, D) for i in range(F)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(ND)]
    m = DeepFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=l2, l2_reg_embedding=l2, dnn_dropout=0, seed=1024,
               device="cpu")
    if tag == "forward_only":
        m.eval()
        with torch.no_grad():
            return timed(lambda: m(X))
    m.compile(opt, "binary_crossentropy", metrics=[])
    m.train()

    def step():
        y_pred = m(X).squeeze()
        m.optim.zero_grad()
        loss = torch.nn.functional.binary_cross_entropy(y_pred, y, reduction="sum")
        total = loss + m.get_regularization_loss() + m.aux_loss
        total.backward()
        m.optim.step()
    return timed(step)


def main():
    import torch
    out = {"host": {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__},
           "workload": "DeepFM Criteo-shaped: 26 x 1M x 16, 13 dense, batch 4096, dnn (256,128)",
           "protocol": "one fresh process per run, 3 warm-up + 5 timed steps", "runs": {}}
    tags = [t for t, _, _ in VARIANTS] + ["forward_only"]
    for which in ("reference", "port"):
        for tag in tags:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", which, tag], capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["runs"][("torch_port_" if which == "port" else "") + tag] = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
    r = out["runs"]
    out["port_over_reference"] = {k: r["torch_port_" + k]["ms_per_step"] / r[k]["ms_per_step"] for k in tags}
    print(json.dumps(out, indent=1))


def cli():
    """bench.py's cpu_baseline leg (kind "reference"): ONE variant of the unmodified reference in this process.
        python oracle/time_reference.py --reference-root oracle/_ref --batch 4096 --vocab 1000000 --steps 10 \
            [--optimizer adagrad --l2 0 | --forward-only] --json
    `--reference-root` is the directory that holds the reference's `deepctr_torch/` package: /root/reference in the build
    container, the unpacked git-ignored archive of oracle/_ref/ (made by __graft_entry__.build()) on the GPU box."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference-root", required=True)
    ap.add_argument("--batch", type=int, default=B)
    ap.add_argument("--vocab", type=int, default=V)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--optimizer", default="adagrad")
    ap.add_argument("--l2", type=float, default=0.0)
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--threads", type=int, default=0, help="torch.set_num_threads (0: torch's default = every logical cpu)")
    a = ap.parse_args()
    import torch
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    sys.path.insert(0, HERE)
    import make_golden as mg
    mg.REFERENCE = os.path.abspath(a.reference_root)
    mg.import_reference()
    from deepctr_torch.inputs import DenseFeat, SparseFeat
    from deepctr_torch.models import DeepFM
    gen = torch.Generator().manual_seed(0)
    X = torch.cat([torch.randint(0, a.vocab, (a.batch, F), generator=gen).float(), torch.rand(a.batch, ND, generator=gen)], 1)
    y = torch.randint(0, 2, (a.batch,), generator=gen).float()
    cols = [SparseFeat("C%d" % (i + 1), a.vocab, D) for i in range(F)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(ND)]
    m = DeepFM(cols, cols, dnn_hidden_units=(256, 128), l2_reg_linear=a.l2, l2_reg_embedding=a.l2, dnn_dropout=0,
               seed=1024, device="cpu")
    if a.forward_only:
        m.eval()
        with torch.no_grad():
            r = timed(lambda: m(X), n=a.steps, warm=3)
    else:
        m.compile(a.optimizer, "binary_crossentropy", metrics=[])
        m.train()

        def step():          # basemodel.py:242-262
            y_pred = m(X).squeeze()
            m.optim.zero_grad()
            loss = torch.nn.functional.binary_cross_entropy(y_pred, y, reduction="sum")
            total = loss + m.get_regularization_loss() + m.aux_loss
            total.backward()
            m.optim.step()
        r = timed(step, n=a.steps, warm=2)
    r.update(value=a.batch / (r["ms_per_step"] * 1e-3), unit="samples/s", steps=a.steps, threads=torch.get_num_threads(),
             reference_root=a.reference_root)
    print(json.dumps(r))


if __name__ == "__main__":
    if "--reference-root" in sys.argv:
        cli()
    elif len(sys.argv) == 4 and sys.argv[1] == "--one":
        print(json.dumps(one(sys.argv[2], sys.argv[3])))
    else:
        main()

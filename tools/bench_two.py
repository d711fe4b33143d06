import os, sys, time, json
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "deepctr-torch_amd"))
import torch
from deepctr_torch.inputs import DenseFeat, SparseFeat
from deepctr_torch import models as M
from deepctr_torch._hip.graph import GraphedTrainStep
dev, B, V = "cuda:0", 4096, 1_000_000
cols = [SparseFeat("C%d" % i, V, 16) for i in range(26)] + [DenseFeat("I%d" % i, 1) for i in range(13)]
gen = torch.Generator().manual_seed(0)
n = B * 8
X = torch.cat([torch.randint(0, V, (n, 26), generator=gen).float(), torch.rand(n, 13, generator=gen)], 1).to(dev)
y = torch.randint(0, 2, (n,), generator=gen).float().to(dev)
for name, make in [("AutoInt", lambda: M.AutoInt(cols, cols, att_layer_num=3, att_head_num=2, dnn_hidden_units=(256, 128), l2_reg_embedding=0, device=dev)),
                   ("AFM", lambda: M.AFM(cols, cols[:26], attention_factor=8, l2_reg_linear=0, l2_reg_embedding=0, l2_reg_att=0, device=dev))]:
    m = make(); m.compile("adagrad", "binary_crossentropy", metrics=[]); m.train()
    batch = lambda i: (X[(i % 8) * B:(i % 8 + 1) * B], y[(i % 8) * B:(i % 8 + 1) * B])
    for i in range(3): m._train_step(*batch(i))
    gs = GraphedTrainStep(m, *batch(0), steps_per_graph=2, inputs_ready=True).capture(*batch(0))
    for i in range(6): gs(*batch(i))
    gs.flush(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(60): gs(*batch(i))
    gs.flush(); torch.cuda.synchronize()
    print(name, "graph ms/step", (time.perf_counter() - t0) / 60 * 1e3)
    del m, gs; torch.cuda.empty_cache()

import sys, os
sys.path.insert(0, "deepctr-torch_amd"); sys.path.insert(0, "tests")
import torch
from test_gpu_update import _model, _batch
DEV="cuda:0"
for idmode in ("uniform", "hot", "same"):
  for B in (64, 300, 4096):
    vocabs=[1000, 17, 100_000, 3]
    X=_batch(B, vocabs, 2, idmode, seed=5)
    res=[]
    for rep in range(3):
        torch.manual_seed(0)
        m=_model(len(vocabs), vocabs, 16, 2)
        m.compile("sgd", "binary_crossentropy")
        gen=torch.Generator(device=DEV).manual_seed(1)
        R=torch.randn(B, m.model_plan().width, device=DEV, generator=gen)
        out, wide, fm = m.fused_inputs(X, want_fm=True)
        o0=out.detach().clone(); f0=fm.detach().clone(); w0=wide.detach().clone()
        ((out*R).sum()+wide.sum()+fm.sum()).backward()
        torch.cuda.synchronize()
        res.append(([p.detach().clone() for p in m.model_plan().table_params], o0, f0, w0))
    for r in res[1:]:
        eq=[torch.equal(a,b) for a,b in zip(res[0][0], r[0])]
        print(idmode, B, "tables equal:", eq, "out", torch.equal(res[0][1], r[1]), "fm", torch.equal(res[0][2], r[2]), "wide", torch.equal(res[0][3], r[3]))
        for a,b in zip(res[0][0], r[0]):
            if not torch.equal(a,b):
                d=(a-b).abs(); idx=d.nonzero()
                print("   ndiff", idx.shape[0], "maxdiff", float(d.max()), "rows", idx[:5].tolist())

import sys, os, ctypes
sys.path.insert(0, "deepctr-torch_amd"); sys.path.insert(0, "tests")
import torch
from test_gpu_update import _model, _batch
from deepctr_torch._hip import lib as L
from deepctr_torch._hip.ops import _ptr
DEV="cuda:0"
lib=L.lib()
B=64
vocabs=[1000, 17]
X=_batch(B, vocabs, 0, "same", seed=5)
m=_model(len(vocabs), vocabs, 16, 0)
plan=m.model_plan(); plan.ensure_gacc(); cplan=plan.bind(DEV)
s=L.stream_handle(DEV)
out=torch.empty(B, plan.ld_out, device=DEV); wide=torch.empty(B, device=DEV); fm=torch.empty(B, device=DEV)
ids_t=torch.empty(len(plan.units), B, dtype=torch.int32, device=DEV); fm_s=torch.empty(B,16,device=DEV)
L.check(lib.dctr_embed_fwd(cplan,_ptr(X),X.stride(0),B,_ptr(out),plan.ld_out,_ptr(wide),_ptr(fm),None,plan.units_ptr(),len(plan.units),_ptr(ids_t),_ptr(fm_s),16,s))
torch.cuda.synchronize()
print("ids_t", ids_t[:, :8].tolist())
gen=torch.Generator(device=DEV).manual_seed(1)
g_out=torch.randn(B, plan.ld_out, device=DEV, generator=gen); g_fm=torch.randn(B,device=DEV,generator=gen); g_w=torch.randn(B,device=DEV,generator=gen)
def run(use_out, use_fm):
    for p in plan.table_params: plan.gacc_of(p).zero_()
    L.check(lib.dctr_embed_update(cplan, plan.units_ptr(), len(plan.units), plan.max_vocab, _ptr(ids_t), B,
        _ptr(g_out) if use_out else None, plan.ld_out, _ptr(out), plan.ld_out, _ptr(fm_s), 16, _ptr(g_fm) if use_fm else None, _ptr(g_w), L.UPD_ACCUM, 0.0, 0.0, None, 0, None, s))
    torch.cuda.synchronize()
    return [plan.gacc_of(p).clone() for p in plan.table_params]
for use_out, use_fm in ((True, False), (False, True), (True, True)):
    rs=[run(use_out,use_fm) for _ in range(4)]
    print("out",use_out,"fm",use_fm,[all(torch.equal(a,b) for a,b in zip(rs[0],r)) for r in rs[1:]])
    # exact expected (sequential descending b order) in torch fp32 on cpu
    G=rs[0][0][999].cpu()
    go=g_out[:, :16].cpu(); e=out[:, :16].cpu(); S=fm_s.cpu(); gf=g_fm.cpu()
    acc=torch.zeros(16)
    for b in range(B-1,-1,-1):
        g=(go[b] if use_out else torch.zeros(16))
        if use_fm: g=g+gf[b]*(S[b]-e[b])
        acc=acc+g
    print("  vs sequential-desc:", float((G-acc).abs().max()))

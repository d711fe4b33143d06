import sys, time
sys.path.insert(0, "deepctr-torch_amd"); sys.path.insert(0, "oracle")
import numpy as np, torch
from deepctr_torch._hip import ops
DEV="cuda:0"
def ref(H, X0, W, b, relu):
    Z = (H[:, :, None, :] * X0[:, None, :, :]).reshape(H.shape[0], -1, H.shape[2])
    Y = torch.einsum("ok,bkd->bod", W, Z) + b[None, :, None]
    return torch.relu(Y) if relu else Y
torch.manual_seed(0)
for (B,h,M,D,O,relu) in [(5,3,3,4,8,1),(33,7,5,16,40,0),(64,26,26,16,128,1),(100,64,26,16,128,1),(17,6,4,8,200,1),(40,2,31,5,32,1)]:
    H=torch.randn(B,h,D,device=DEV)*0.5; X0=torch.randn(B,M,D,device=DEV)*0.5; W=torch.randn(O,h*M,device=DEV)*0.1; b=torch.randn(O,device=DEV)*0.1
    A=ops.cin_layer_forward(H,X0,W,b,relu)
    R=ref(H.double(),X0.double(),W.double(),b.double(),relu)
    err=float((A.double()-R).abs().max()); print((B,h,M,D,O,relu),"max err %.3e"%err, "scale %.2f"%float(R.abs().max()))
# strided views
B=64; out=torch.randn(B, 26*16+13+3, device=DEV); X0=out[:, :416].reshape(B,26,16)
Aprev=torch.randn(B,128,16,device=DEV); H=Aprev[:, :64]
W=torch.randn(128,64*26,device=DEV)*0.05; b=torch.zeros(128,device=DEV)
A=ops.cin_layer_forward(H,X0,W,b,1); R=ref(H.double(),X0.double(),W.double(),b.double(),1)
print("strided", float((A.double()-R).abs().max()))
# timing at the bench shape
B=4096; X0=torch.randn(B,26,16,device=DEV)*0.1
for h in (26,64):
    H=torch.randn(B,h,16,device=DEV)*0.1; W=torch.randn(128,h*26,device=DEV)*0.05; b=torch.zeros(128,device=DEV)
    for _ in range(3): ops.cin_layer_forward(H,X0,W,b,1)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.cin_layer_forward(H,X0,W,b,1)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*100; fl=2*B*16*128*h*26
    print("h=%d: %.1f us/layer, %.1f TFLOP/s"%(h,us,fl/us/1e6))
    t0=time.perf_counter()
    Z=(H[:, :, None, :] * X0[:, None, :, :]).reshape(B,-1,16); Y=torch.relu(torch.einsum("ok,bkd->bod",W,Z)); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        Z=(H[:, :, None, :] * X0[:, None, :, :]).reshape(B,-1,16); Y=torch.relu(torch.einsum("ok,bkd->bod",W,Z))
    e1.record(); torch.cuda.synchronize(); print("   torch einsum path: %.1f us"%(e0.elapsed_time(e1)*200))

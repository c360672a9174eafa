import os, sys
sys.path.insert(0, os.getcwd())
import torch
from maskrcnn_amd import ops
dev="cuda:0"; g=torch.Generator().manual_seed(0)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr=torch.cuda.CUDAGraph(); s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(iters): fn()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/(3*iters)
for name,(b,h,w,cin,cout,k) in {"rpn_p2":(8,208,336,256,512,3),"fpn_p2":(8,208,336,256,256,3),"c4_conv2":(8,52,84,256,256,3)}.items():
    fl=2.0*b*h*w*cin*cout*k*k
    for fill in ("random","zeros","ones"):
        if fill=="random":
            x=torch.randn(b,h,w,cin,generator=g).half().to(dev); wt=(torch.randn(cout,k,k,cin,generator=g)*0.02).half().to(dev)
        elif fill=="zeros":
            x=torch.zeros(b,h,w,cin,dtype=torch.float16,device=dev); wt=torch.zeros(cout,k,k,cin,dtype=torch.float16,device=dev)
        else:
            x=torch.ones(b,h,w,cin,dtype=torch.float16,device=dev); wt=torch.full((cout,k,k,cin),0.001,dtype=torch.float16,device=dev)
        ms=timeit(lambda: ops.conv_f16_pipelined(x,wt,None,None,pad=(1,1,1,1),relu=True))
        print(name, fill, round(ms*1e3,1), "us", round(fl/ms/1e9,1), "TFLOP/s", flush=True)

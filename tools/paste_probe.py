"""Tuning aid for mrcnn_paste_masks_u8: time for 400 identical boxes of a given size on a 1024^2 canvas (exposes
load imbalance and per-workgroup latency, which a random box mix hides), and single-detection latency."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from maskrcnn_amd import ops
from maskrcnn_amd._lib import lib, check
dev='cuda:0'
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
rng=np.random.default_rng(0)
n,c=400,81
masks=torch.from_numpy((1/(1+np.exp(-rng.normal(0,3,(n,28,28,c))))).astype(np.float32)).to(dev)
ids=torch.from_numpy(rng.integers(1,c,n).astype(np.int64)).to(dev)
out=torch.empty(n,1024,1024,dtype=torch.uint8,device=dev)
def run(boxes):
    b=torch.tensor(boxes,dtype=torch.float32,device=dev)
    s=torch.cuda.current_stream().cuda_stream
    sn,sy,sx,sc=masks.stride()
    return timeit(lambda: check(lib.mrcnn_paste_masks_u8(masks.data_ptr(),sn,sy,sx,sc,n,28,28,c,ids.data_ptr(),b.data_ptr(),1024,1024,1,out.data_ptr(),s)))
print('memset only ~', timeit(lambda: out.zero_()))
for name,box in (('empty',[0,0,0,0]),('1x1',[5,5,6,6]),('32x32',[100,100,132,132]),('160x160',[100,100,260,260]),('28x600',[100,100,128,700]),('600x28',[100,100,700,128]),('600x600',[100,100,700,700]),('full',[0,0,1024,1024])):
    print(name, round(run([box]*n),1),'us')
print('--- longer runs (clock ramp?)')
for iters in (30, 300, 1500):
    b=torch.tensor([[100,100,132,132]]*n,dtype=torch.float32,device=dev)
    s=torch.cuda.current_stream().cuda_stream
    sn,sy,sx,sc=masks.stride()
    t=timeit(lambda: check(lib.mrcnn_paste_masks_u8(masks.data_ptr(),sn,sy,sx,sc,n,28,28,c,ids.data_ptr(),b.data_ptr(),1024,1024,1,out.data_ptr(),s)), iters=iters)
    print(iters, round(t,1))
# one detection only: single WG latency
b=torch.tensor([[100,100,132,132]],dtype=torch.float32,device=dev)
t=timeit(lambda: check(lib.mrcnn_paste_masks_u8(masks.data_ptr(),sn,sy,sx,sc,1,28,28,c,ids.data_ptr(),b.data_ptr(),1024,1024,1,out.data_ptr(),s)), iters=200)
print('n=1 32x32', round(t,1))
b=torch.tensor([[0,0,0,0]],dtype=torch.float32,device=dev)
t=timeit(lambda: check(lib.mrcnn_paste_masks_u8(masks.data_ptr(),sn,sy,sx,sc,1,28,28,c,ids.data_ptr(),b.data_ptr(),1024,1024,1,out.data_ptr(),s)), iters=200)
print('n=1 empty', round(t,1))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
dbg = torch.zeros(64 * 64, dtype=torch.int64, device=dev)
os.environ["MRCNN_CONV_DBG"] = str(dbg.data_ptr())
from maskrcnn_amd import ops
x = torch.randn(8, 256, 256, 256, device=dev); w = torch.randn(512, 3, 3, 256, device=dev) * 0.05
for _ in range(2): ops.conv_bn_act(x, w, None, None, 1, (1, 1, 1, 1))
torch.cuda.synchronize()
d = dbg.cpu().view(64, 8, 8)
import numpy as np
a = d.numpy().astype(np.int64)
# per k-tile intervals: 0->1 (chunks 0-2 issue), 1->2 (vmcnt wait + ds_write), 2->3 (barrier), 3->4 (loads+read issue), 4->next 0 (chunk 3 issue)
rows = []
for b in range(64):
    for k in range(7):
        t = a[b, k]; n0 = a[b, k + 1, 0]
        rows.append([t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], n0-t[4], n0-t[0]])
r = np.array(rows)
print("median cycles: chunks0-2 %d | wait+ds_write %d | barrier %d | loads+frag issue %d | chunk3 %d | total k-tile %d" % tuple(np.median(r, 0)))
print("mean   cycles: chunks0-2 %d | wait+ds_write %d | barrier %d | loads+frag issue %d | chunk3 %d | total k-tile %d" % tuple(r.mean(0)))

import sys, torch
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[R, R+'/puzzlefusion-plusplus_amd']
from pfpp_hip import ops
dev=torch.device('cuda:0')
for G,S in ((19712,128),(3850,25),(40000,125)):
    F=G//S
    idx=torch.randint(0,256,(F,S,64),dtype=torch.int32,device=dev)
    half=torch.rand(F,S,device=dev)<0.5
    idx[half][:, 32:]=0
    idx2=idx.clone(); idx2[half]=torch.where(torch.arange(64,device=dev)>=32, idx2[half][:, :1], idx2[half])
    s=ops.sa_pad_schedule(idx2).cpu(); torch.cuda.synchronize()
    two=(idx2[:,:,32:]!=idx2[:,:,:1]).any(-1).reshape(-1).cpu()
    n2=int(s[F*S]); import numpy as np
    ok = n2==int(two.sum()) and np.array_equal(s[:n2].numpy(), np.nonzero(two.numpy())[0]) and np.array_equal(s[n2:F*S].numpy(), np.nonzero(~two.numpy())[0])
    for _ in range(5): ops.sa_pad_schedule(idx2)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ops.sa_pad_schedule(idx2)
    e1.record(); torch.cuda.synchronize()
    print(G, 'correct', ok, 'n2', n2, f'{e0.elapsed_time(e1)*10:.1f} us per call (2 launches + alloc)')

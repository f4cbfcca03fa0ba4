import os,sys,json,statistics,torch
sys.path.insert(0,"/root/repo" if os.path.isdir("/root/repo/deqsci_amd") else ".")
from deqsci_amd import _hip
_hip._LIB_PATH=os.path.abspath(os.environ["W44_LIB"])
g=torch.Generator(device="cuda").manual_seed(5)
w=torch.randn(64,64,3,3,device="cuda",generator=g)*0.05; b=torch.randn(64,device="cuda",generator=g)
shape=(64,128,128)
x=torch.randn(shape[0],64,shape[1],shape[2],device="cuda",generator=g).contiguous(memory_format=torch.channels_last)
U=_hip.pack_winograd44_weights(w); xb=_hip.Blk32.from_nchw(x); ob=_hip.Blk32.empty(*shape,"cuda"); on=torch.empty_like(x)
res={}
for name,fn in (("nhwc->nhwc",lambda: _hip.conv3x3_c64_winograd44(x,U,b,True,out=on)),("blk->blk",lambda: _hip.conv3x3_c64_winograd44(xb,U,b,True,out=ob,out_blk=True))):
    ts=[]
    for r in range(7):
        for _ in range(3): fn()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/20*1e3)
    res[name]=round(statistics.median(ts),1)
print(json.dumps({"lib":os.environ["W44_LIB"],**res}))

import os,sys,json,statistics,ctypes,torch
sys.path.insert(0,".")
from deqsci_amd import _hip
_hip._LIB_PATH=os.path.abspath(os.environ["W44_LIB"])
_hip.SIGNATURES.pop("deqsci_conv3x3_c64_winograd44_layout_f32", None)
lib=_hip.load()
g=torch.Generator(device="cuda").manual_seed(5)
w=torch.randn(64,64,3,3,device="cuda",generator=g)*0.05; b=torch.randn(64,device="cuda",generator=g)
x=torch.randn(64,64,128,128,device="cuda",generator=g).contiguous(memory_format=torch.channels_last); out=torch.empty_like(x)
U=_hip.pack_winograd44_weights(w)
def fn():
    rc=lib.deqsci_conv3x3_c64_winograd44_f32(x.data_ptr(),U.data_ptr(),b.data_ptr(),out.data_ptr(),64,128,128,1,torch.cuda.current_stream().cuda_stream)
    assert rc==0
ts=[]
for r in range(7):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/20*1e3)
print(json.dumps({"lib":os.environ["W44_LIB"],"us":round(statistics.median(ts),1)}))

"""Phase timing of the Winograd kernel from s_memtime stamps (library built with -DWG_STAMP: the `bias` pointer is
re-purposed as the stamp buffer, 40 uint64 per block)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deqsci_amd import _hip
N, H, W = 64, 128, 128
x = torch.randn(N, 64, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
U = _hip.pack_winograd_weights(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
nblk = (W // 16) * (H // 16) * N
stamps = torch.zeros(nblk * 40, dtype=torch.int64, device="cuda")
out = torch.empty_like(x)
for _ in range(3):
    _hip.load().deqsci_conv3x3_c64_winograd_f32(x.data_ptr(), U.data_ptr(), stamps.data_ptr(), out.data_ptr(), N, H, W, 0, None)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nblk, 40)[:, :27].astype(np.int64)
d = np.diff(s, axis=1)
names = ["prologue"] + sum([[f"c{c} Vcompute", f"c{c} MFMA phase", f"c{c} store+barrier"] for c in range(8)], []) + ["epilogue"]
med = np.median(d, axis=0)
tot = np.median(s[:, 26] - s[:, 0])
print("median block lifetime (s_memtime ticks):", tot)
agg = {"prologue": med[0], "V compute (8)": med[1:25:3].sum(), "MFMA phases (8)": med[2:25:3].sum(), "store_u/raw + barriers (8)": med[3:25:3].sum(), "epilogue": med[25]}
for k, v in agg.items():
    print(f"  {k:28s} {v:9.0f}  {100 * v / tot:5.1f} %")
print("  per chunk MFMA phase median:", med[2:25:3])

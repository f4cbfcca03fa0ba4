"""Phase timing of the Winograd kernel from s_memtime stamps (library built with -DWG_STAMP: the `bias` pointer is
re-purposed as the stamp buffer, 40 uint64 per block; slot 38/39 = XCC_ID / HW_ID of wave 0)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deqsci_amd import _hip
if os.environ.get("WG_LIB"):                                  # a variant library built by tools/wg_variants.sh
    _hip._LIB_PATH = os.environ["WG_LIB"]
N, H, W = 64, 128, 128
x = torch.randn(N, 64, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
U = _hip.pack_winograd_weights(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
cap = 4 * (W // 16) * (H // 16) * N                         # enough for 16-tile blocks
stamps = torch.zeros(cap * 40, dtype=torch.int64, device="cuda")
out = torch.empty_like(x)
for _ in range(3):
    stamps.zero_()
    _hip.load().deqsci_conv3x3_c64_winograd_f32(x.data_ptr(), U.data_ptr(), stamps.data_ptr(), out.data_ptr(), N, H, W, 0, None)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(cap, 40).astype(np.int64)
s = s[s[:, 0] != 0]
nblk = len(s)
nz = int((s[0, :38] != 0).sum())
d = np.diff(s[:, :nz], axis=1)
med = np.median(d, axis=0)
print("blocks:", nblk, " stamps per block:", nz, " median block lifetime:", np.median(s[:, nz - 1] - s[:, 0]))
print("median deltas:", [int(v) for v in med])
# co-residency: group blocks by (xcc, se, sh, cu); within a CU, what fraction of the busy span has >= 1 block inside an
# MFMA phase (marks 3k+2 -> 3k+3 for chunk k in the one-loop variants), and how often 0 / 1 / 2 blocks are in one
hw, xcc = s[:, 39], s[:, 38] & 0xF
cu_key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 0xF)
keys = np.unique(cu_key)
print("distinct CUs seen:", len(keys))
if "--timeline" in sys.argv:
    k0 = keys[len(keys) // 2]
    blk = s[cu_key == k0]
    blk = blk[np.argsort(blk[:, 0])]
    t0 = blk[0, 0]
    for b in blk[:8]:
        print("simd/wave %d/%d" % ((b[39] >> 4) & 3, b[39] & 0xF), [int(v - t0) for v in b[:nz]])
        ph = b[2:nz - 1]
        print("   phases (mfma, other+barrier, gap):", [(int(ph[i + 1] - ph[i]), int(ph[i + 2] - ph[i + 1]), int(ph[i + 3] - ph[i + 2])) for i in range(0, len(ph) - 3, 3)])
mf0 = int(os.environ.get("WG_MF0", 2)); per = int(os.environ.get("WG_PER", 3))
tot = np.zeros(3); span = 0
for k0 in keys:
    blk = s[cu_key == k0]
    ev = []
    for b in blk:
        for c in range(8):
            ev.append((b[mf0 + per * c], 1)); ev.append((b[mf0 + per * c + 1], -1))
    ev.sort()
    lvl, last = 0, blk[:, 0].min()
    end = blk[:, nz - 1].max()
    for t, dl in ev:
        tot[min(lvl, 2)] += t - last; last = t; lvl += dl
    tot[0] += end - last
print("fraction of CU time with 0 / 1 / 2 blocks in an MFMA phase:", np.round(tot / tot.sum(), 3))

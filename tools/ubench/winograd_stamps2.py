"""Fine timeline of the two waves of one SIMD (waves 0 and 4 of a workgroup) of the Winograd kernel, from s_memtime stamps
(variant library built with -DWG_STAMP -DWG_STAMP_SKIP=n, tools/wg_variants.sh; WG_LIB points at it).  Nine marks per
pipeline stage: start | raw DMA issued | operands preloaded | after xi 3 | 7 | 11 | 15 (MFMA phase done) | before barrier |
after barrier.  Prints, for the median block, every stage of both waves as offsets from the stage's first mark of wave 0."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deqsci_amd import _hip
if os.environ.get("WG_LIB"):
    _hip._LIB_PATH = os.environ["WG_LIB"]
N, H, W = (int(v) for v in os.environ.get("WG_SHAPE", "64,128,128").split(","))
skip = int(os.environ.get("WG_STAMP_SKIP", "119"))
x = torch.randn(N, 64, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
U = _hip.pack_winograd_weights(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
nblk = 256
stamps = torch.zeros(nblk * 96, dtype=torch.int64, device="cuda")
out = torch.empty_like(x)
for _ in range(3):
    stamps.zero_()
    _hip.load().deqsci_conv3x3_c64_winograd_f32(x.data_ptr(), U.data_ptr(), stamps.data_ptr(), out.data_ptr(), N, H, W, 0, None)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nblk, 2, 48).astype(np.int64)
ok = s[:, 0, 0] != 0
s = s[ok]
life = s[:, 0, 45] - s[:, 0, 0]
b = s[np.argsort(life)[len(life) // 2]]
print("blocks:", len(s), "median span of the 46 marks:", int(np.median(life)), "simd of wave0/wave4:", (b[0, 47] >> 4) & 3, (b[1, 47] >> 4) & 3)
first_stage_mark = (2 - skip) % 9                        # index in the window of the first 'stage start' mark
names = ["start", "dma", "pre", "xi3", "xi7", "xi11", "xi15", "preB", "postB"]
i = first_stage_mark
stage_no = (skip + i - 2) // 9
while i + 9 <= 46:
    t0 = b[0, i]
    print(f"stage {stage_no % 8} (tile {stage_no // 8}): span {int(b[0, i + 9] - t0) if i + 9 < 46 else -1}")
    for wv in (0, 1):
        print("   wave", 4 * wv, " ".join(f"{names[k]}={int(b[wv, i + k] - t0):5d}" for k in range(9)))
    i += 9
    stage_no += 1
# aggregate over blocks: mean duration of each segment per wave, for stages that are not tile boundaries
seg = np.zeros((2, 9)); cnt = 0
i = first_stage_mark
stage_no = (skip + i - 2) // 9
while i + 10 <= 46:
    if stage_no % 8 not in (7, 0):
        for wv in (0, 1):
            seg[wv] += np.median(np.diff(s[:, wv, i:i + 10], axis=1), axis=0)
        cnt += 1
    i += 9
    stage_no += 1
if cnt:
    for wv in (0, 1):
        print("median segment lengths, wave", 4 * wv, [int(v) for v in seg[wv] / cnt], "(start->dma, dma->pre, pre->xi3, ->xi7, ->xi11, ->xi15, ->preB, ->postB, ->next start)")

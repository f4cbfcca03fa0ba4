import sys, random, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deqsci_amd import _hip
random.seed(7)
g = torch.Generator(device="cuda").manual_seed(7)
bad = 0
shapes = [(1, 1, 33), (1, 33, 1), (2, 15, 31), (1, 16, 33), (3, 31, 97), (5, 47, 65), (1, 300, 40), (2, 7, 260), (9, 64, 64), (257, 16, 32), (1, 129, 129)]
shapes += [(random.randint(1, 6), random.randint(1, 90), random.randint(1, 140)) for _ in range(14)]
for (n, H, W) in shapes:
    x = torch.randn(n, 64, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g)
    U = _hip.pack_winograd44_weights(w)
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    outs = {"nhwc": _hip.conv3x3_c64_winograd44(x, U, b, True),
            "blk_in": _hip.conv3x3_c64_winograd44(_hip.Blk32.from_nchw(x), U, b, True),
            "blk_out": _hip.conv3x3_c64_winograd44(x, U, b, True, out_blk=True).to_nchw(),
            "blk_both": _hip.conv3x3_c64_winograd44(_hip.Blk32.from_nchw(x), U, b, True, out_blk=True).to_nchw()}
    errs = {k: float((v.double() - want).norm() / want.norm()) for k, v in outs.items()}
    ok = all(e < 4e-6 for e in errs.values()) and all(torch.equal(outs["nhwc"], v) for v in outs.values())
    bad += not ok
    print((n, H, W), "ok" if ok else "BAD", {k: "%.2e" % e for k, e in errs.items()})
print("fuzz:", "FAILED %d" % bad if bad else "all ok")

#!/usr/bin/env python3
"""anderson_solve: the shipped library against another build of it (default build/wgv/lib_r1.so = round 1's serial form) on
random histories - alpha / residuals / Gram must agree bit for bit - and the launch time of each."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip

new = _hip.load()
old = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "wgv", "lib_r1.so"))
for lib in (old,):
    lib.deqsci_anderson_solve_f32.argtypes = _hip.SIGNATURES["deqsci_anderson_solve_f32"]
    lib.deqsci_anderson_solve_f32.restype = ctypes.c_int
st = lambda: torch.cuda.current_stream().cuda_stream
ok = True
for bsz, N, m in ((1, 524288, 5), (8, 524288, 5), (3, 4096 + 12, 3), (2, 16 * 16 * 8, 8)):
    nch = new.deqsci_anderson_chunks(bsz, N)
    g = torch.Generator(device="cuda").manual_seed(bsz * 7 + m)
    grams = [torch.zeros(new.deqsci_gram_bytes(bsz) // 8, device="cuda", dtype=torch.float64) for _ in range(2)]
    alphas = [torch.zeros(bsz, 8, device="cuda") for _ in range(2)]
    ress = [torch.zeros(1 + bsz, device="cuda") for _ in range(2)]
    for k in range(3 * m):
        slot, nf = k % m, min(k + 1, m)
        n = 0 if k == 0 else nf
        part = torch.zeros(bsz, nch, 9, device="cuda")
        base = torch.randn(bsz, nch, 9, device="cuda", generator=g)
        part[..., :nf] = base[..., :nf] * 0.01
        part[..., slot] = base[..., slot].abs() + 0.5            # diagonal-ish dominance is not required; keep it generic
        part[..., 8] = base[..., 8].abs() + 1.0
        for i, lib in enumerate((new, old)):
            rc = lib.deqsci_anderson_solve_f32(part.data_ptr(), grams[i].data_ptr(), alphas[i].data_ptr(), ress[i].data_ptr(), bsz, N, m, slot, nf, n,
                                               1e-2, 1e-5, st())
            assert rc == 0, rc
        torch.cuda.synchronize()
        same = torch.equal(alphas[0], alphas[1]) and torch.equal(ress[0], ress[1]) and torch.equal(grams[0][:bsz * 80], grams[1][:bsz * 80])
        if not same:
            ok = False
            print("MISMATCH", bsz, N, m, k, (alphas[0] - alphas[1]).abs().max().item(), (ress[0] - ress[1]).abs().max().item())
    times = []
    for i, lib in enumerate((new, old)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            lib.deqsci_anderson_solve_f32(part.data_ptr(), grams[i].data_ptr(), alphas[i].data_ptr(), ress[i].data_ptr(), bsz, N, m, slot, nf, n, 1e-2, 1e-5, st())
        e0.record()
        for _ in range(50):
            lib.deqsci_anderson_solve_f32(part.data_ptr(), grams[i].data_ptr(), alphas[i].data_ptr(), ress[i].data_ptr(), bsz, N, m, slot, nf, n, 1e-2, 1e-5, st())
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f'{{"bsz": {bsz}, "N": {N}, "m": {m}, "chunks": {nch}, "new_us": {times[0]:.1f}, "round1_us": {times[1]:.1f}, "bit_identical": {str(ok).lower()}}}')
sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""Socket power and shader clock while the split-fp16 conv kernel (or a variant library, DEQSCI_HIP_LIB) runs back to back for a few seconds:
is the launch time energy / power?  Samples the amdgpu hwmon files (power1_average / power1_input in uW, freq1_input in Hz) from a thread;
prints what it finds when the files are missing (then `rocm-smi` is tried once per phase)."""
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip  # noqa: E402


def hwmon_files():
    """hwmon files of the GPU this process computes on (a box shows every card of its node in sysfs): matched by PCI address."""
    want = None
    try:
        pr = torch.cuda.get_device_properties(0)
        want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:  # noqa: BLE001
        pass
    cands = []
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        addr = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
        files = {name: os.path.join(d, name) for name in ("power1_average", "power1_input", "freq1_input", "power1_cap") if os.path.exists(os.path.join(d, name))}
        if files:
            cands.append((addr == want, addr, files))
    cands.sort(key=lambda c: not c[0])
    if not cands:
        return {}
    out = dict(cands[0][2])
    out["_pci"] = cands[0][1] + ("" if cands[0][0] else " (no PCI match for %s: first card)" % want)
    return out


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.rows, self.stop = files, [], False

    def run(self):
        while not self.stop:
            row = {}
            for k, f in self.files.items():
                try:
                    row[k] = int(open(f).read().strip())
                except (OSError, ValueError):
                    pass
            self.rows.append(row)
            time.sleep(0.02)


def main():
    files = hwmon_files()
    print(json.dumps({"hwmon": files}))
    pci = files.pop("_pci", None)
    zero = bool(os.environ.get("S16_ZERO"))
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g)
    NI = int(os.environ.get("PROBE_IMAGES", "64"))             # 64 = the bench shape (16 block tiles per CU); 8 = one measurement per call (ONE tile per CU)
    x = torch.randn(NI, 64, 128, 128, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    if zero:
        x.zero_(); w.zero_()
    kind = os.environ.get("PROBE_KERNEL", "s16")               # s16 | stack | w16 | w16stack | f44 (Winograd F(4x4,3x3), blk32 -> blk32) | f22 (Winograd F(2x2,3x3))
    if kind == "s16":
        xs = _hip.to_split16(x); out = _hip.Sp16.empty(NI, 128, 128, "cuda"); Wsp = _hip.Split16Weights(w)
        launch = lambda: _hip.conv3x3_c64_split16(xs, Wsp, b, True, out=out)  # noqa: E731
    elif kind == "stack":                                      # 13 such layers as ONE launch (PROBE_IMAGES=8: the stack launch's shape); per-launch figures = 13 layers
        xs = _hip.to_split16(x)
        stack = _hip.Split16Stack([(_hip.Split16Weights(w), b, True)] * 13, "cuda")
        launch = lambda: _hip.conv3x3_c64_split16_stack(xs, stack, check=False)  # noqa: E731
    elif kind == "w16stack":                                   # ... on the Winograd kernel (csrc/conv_w16.hip)
        xp = _hip.P32.from_nchw(x)
        stack = _hip.Wino16Stack([(_hip.Wino16Weights(w), b, True)] * 13, "cuda")
        launch = lambda: _hip.conv3x3_c64_wino16_stack(xp, stack, check=False)  # noqa: E731
    elif kind == "w16":
        xp = _hip.P32.from_nchw(x); op = _hip.P32.empty(NI, 128, 128, "cuda"); Ww = _hip.Wino16Weights(w)
        launch = lambda: _hip.conv3x3_c64_wino16(xp, Ww, b, True, out=op)  # noqa: E731
    elif kind == "f44":
        xb = _hip.Blk32.from_nchw(x); ob = _hip.Blk32.empty(NI, 128, 128, "cuda"); U = _hip.pack_winograd44_weights(w)
        launch = lambda: _hip.conv3x3_c64_winograd44(xb, U, b, True, out=ob, out_blk=True)  # noqa: E731
    else:
        o2 = torch.empty_like(x); U2 = _hip.pack_winograd_weights(w)
        launch = lambda: _hip.conv3x3_c64_winograd(x, U2, b, True, out=o2)  # noqa: E731
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    time.sleep(1.0)
    idle = {k: int(open(f).read().strip()) for k, f in files.items() if k != "power1_cap"} if files else {}
    s = Sampler({k: f for k, f in files.items() if k != "power1_cap"})
    s.start()
    n = (12000 if kind == "s16" else 8000) * max(1, 64 // NI) // (1 if NI == 64 else 2) // (13 if kind in ("stack", "w16stack") else 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    s.stop = True
    us = e0.elapsed_time(e1) / n * 1e3
    rows = s.rows[len(s.rows) // 4:]                     # steady part
    res = {"kernel": kind, "images": NI, "lib": os.environ.get("DEQSCI_HIP_LIB", "product"), "zero_operands": zero, "launch_us": round(us, 1), "samples": len(rows), "idle": idle, "pci": pci}
    for k in ("power1_average", "power1_input", "freq1_input"):
        v = [r[k] for r in rows if k in r]
        if v:
            res[k + "_median"] = statistics.median(v)
    if "power1_cap" in files:
        res["power1_cap"] = int(open(files["power1_cap"]).read().strip())
    p = res.get("power1_average_median") or res.get("power1_input_median")
    if p:
        res["energy_per_launch_mJ"] = round(p * 1e-6 * us * 1e-6 * 1e3, 2)
    if not files:
        try:
            res["rocm_smi"] = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=30).stdout[-1500:]
        except Exception as e:  # noqa: BLE001
            res["rocm_smi"] = repr(e)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

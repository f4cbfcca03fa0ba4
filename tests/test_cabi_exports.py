"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/deqsci_hip.h
declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "deqsci_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(deqsci_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from deqsci_amd import _hip
    assert os.path.exists(_hip.lib_path()), "run `make` / __graft_entry__.build() first"
    lib = ctypes.CDLL(_hip.lib_path())
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/deqsci_hip.h but not exported"
    assert set(_hip.SIGNATURES) | set(_hip.OTHER_EXPORTS) == set(syms)


def test_scratch_size_queries_and_error_strings():
    from deqsci_amd import _hip
    lib = _hip.load()
    assert b"gfx950" in lib.deqsci_version()
    n = lib.deqsci_anderson_chunks(8, 256 * 256 * 8)
    assert n > 0 and lib.deqsci_partials_bytes(8, 256 * 256 * 8) == 8 * n * _hip.PART_STRIDE * 4
    assert lib.deqsci_gram_bytes(3) == (3 * 80 + 2) * 8
    assert lib.deqsci_anderson_chunks(0, 10) == 0
    assert lib.deqsci_conv3x3_c64_split16_stack(None, None, None, None, 13, 8, 128, 128, None, 8, 8, 8, None, None, None, None) == -1
    for code in (-1, -2, -3, -4):
        assert len(lib.deqsci_error_string(code)) > 5
    # argument validation happens before any launch, so it is testable without a GPU
    assert lib.deqsci_sci_forward_f32(None, None, None, 1, 4, 4, 8, 0, 0, None) == -1
    assert lib.deqsci_residual_store_f32(None, None, None, None, None, None, None, 1, 64, 5, 0, 1, None) == -1
    # the reference's Gram arithmetic (round 5): state size = per sample the 8 x 8 Gram, chain sums, counters, block records and term slots
    nb = lib.deqsci_gram_ref_bytes(8, 256 * 256 * 8)
    assert nb % 16 == 0 and nb // 8 > 4 * (64 + 128 + n * 8 * 16 * 2 * 2) and lib.deqsci_gram_ref_bytes(0, 10) == 0
    assert lib.deqsci_anderson_solve_ref_f32(None, None, None, None, None, None, 1, 64, 5, 0, 1, 0, 1e-2, 1e-5, None) == -1
    assert lib.deqsci_gram_row_chain16_f32(None, None, None, 1, 64, 5, 0, 1, 0, None) == -1
    # round 6: K4 fused with the reference Gram's first pass - where the blocks are 2048 elements and N is a whole number of them
    assert lib.deqsci_gram_ref_fusable(8, 256 * 256 * 8) == 1 and lib.deqsci_gram_ref_fusable(2, 512 * 512 * 16) == 0      # (2048 blocks per sample: the two launches)
    assert lib.deqsci_gram_ref_fusable(1, 2048) == 1 and lib.deqsci_gram_ref_fusable(1, 2048 + 4) == 0 and lib.deqsci_gram_ref_fusable(64, 1 << 22) == 0
    assert lib.deqsci_gram_ref_fusable(0, 2048) == 0
    assert lib.deqsci_residual_store_ref_f32(None, None, None, None, None, None, None, None, 1, 2048, 5, 0, 1, None) == -1
    assert lib.deqsci_anderson_apply_solve_ref_f32(None, None, None, None, None, None, 1, 2048, 5, 0, 1, 0, 1e-2, 1e-5, None) == -1
    buf = (ctypes.c_float * 96)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) // 16 * 16
    assert lib.deqsci_sci_forward_f32(p16, p16, p16, 1, 2, 2, -8, 0, 0, None) == -2
    assert lib.deqsci_sci_forward_f32(p16 + 4, p16, p16, 1, 2, 2, 8, 0, 0, None) == -3
    assert lib.deqsci_gram_row_chain16_f32(p16, p16, p16, 1, 64, 5, 5, 5, 0, None) == -2           # slot >= m
    assert lib.deqsci_gram_row_chain16_f32(p16, p16, p16 + 4, 1, 64, 5, 0, 1, 0, None) == -3
    assert lib.deqsci_gap_update_f32(p16, p16, p16, p16, p16, 1, 2, 2, 8, 0, 1, 0, None) == -4     # aliasing across layouts
    assert lib.deqsci_anderson_mix_f32(p16, p16, p16, p16, 1.0, 9, 1, 8, 9, None) == -4           # m > DEQSCI_MAX_M
    # Winograd conv: the kernel forms a 32-bit byte offset (pixel * 256 B), so images of 2^24 pixels or more are refused
    q16 = p16 + 64
    assert lib.deqsci_conv3x3_c64_winograd_f32(p16, p16, None, q16, 1, 4096, 4096, 1, None) == -4
    assert lib.deqsci_conv3x3_c64_winograd_f32(p16, p16, None, q16, 1, 8192, 2048, 1, None) == -4
    assert lib.deqsci_conv3x3_c64_winograd_f32(p16, p16, None, p16, 1, 16, 16, 1, None) == -4      # in place
    assert lib.deqsci_conv3x3_c64_winograd_f32(p16, None, None, q16, 1, 16, 16, 1, None) == -1
    assert lib.deqsci_conv3x3_c64_winograd_f32(p16 + 4, p16, None, q16, 1, 16, 16, 1, None) == -3
    # the F(4x4,3x3) entry point has the same contract
    assert lib.deqsci_conv3x3_c64_winograd44_f32(p16, p16, None, q16, 1, 4096, 4096, 1, None) == -4
    # ... and a tighter size limit: its zero padding needs the out-of-range sentinel offset 2^31 to lie beyond the image
    assert lib.deqsci_conv3x3_c64_winograd44_f32(p16, p16, None, q16, 1, 2900, 2900, 1, None) == -4
    assert lib.deqsci_conv3x3_c64_winograd44_f32(p16, p16, None, q16, 1, 8192, 1025, 1, None) == -4   # 1025 columns pad to 1056
    assert _hip.conv64_kernel_for(64, 2900, 2900, policy="fast") == "f22"                           # the front end falls back by itself
    assert _hip.W44_MAX_PIXELS * 256 + 4096 + 2048 + 16 <= 2 ** 31 < (_hip.W44_MAX_PIXELS + 1) * 256 + 4096 + 2048 + 16
    # the split-fp16 launcher's margin is the stricter one: an image between the two limits is routed to F(2x2,3x3), not to a launcher that refuses
    assert _hip.S16_MAX_PIXELS * 256 + 4096 + 4096 + 16 <= 2 ** 31 < (_hip.S16_MAX_PIXELS + 1) * 256 + 4096 + 4096 + 16 and _hip.S16_MAX_PIXELS < _hip.W44_MAX_PIXELS
    hh_h, hh_w = 511, 32 * 513                                                                      # 8388576 pixels (the advisor's example), 32-column aligned
    assert _hip.S16_MAX_PIXELS < hh_h * hh_w <= _hip.W44_MAX_PIXELS
    assert _hip.conv64_kernel_for(64, hh_h, hh_w, policy="fast") == "f22"
    assert lib.deqsci_conv3x3_c64_split16(p16, p16, None, q16, 1, hh_h, hh_w, 1, 0, None, 8, None, 8, None, 0, None, None, None) == -4
    assert lib.deqsci_conv3x3_c64_winograd44_f32(p16, p16, None, p16, 1, 16, 16, 1, None) == -4    # in place
    assert lib.deqsci_conv3x3_c64_winograd44_f32(p16, None, None, q16, 1, 16, 16, 1, None) == -1
    assert lib.deqsci_conv3x3_c64_winograd44_f32(p16 + 4, p16, None, q16, 1, 16, 16, 1, None) == -3
    assert lib.deqsci_conv3x3_c64_winograd44_f32(p16, p16, None, q16, 0, 16, 16, 1, None) == -2
    assert lib.deqsci_conv3x3_c64_winograd44_layout_f32(p16, p16, None, q16, 1, 16, 16, 1, 2, 0, None, None, None) == -4   # unknown layout
    assert lib.deqsci_conv3x3_c64_winograd44_layout_f32(p16, p16, None, q16, 1, 16, 16, 1, 1, 1, None, p16, None) == -1   # one event only
    # the split-fp16 convolution and its edge layers: the same contract

    def s16(x, w, y, n=1, H=16, W=16, w_exp=0, in_exp=8, out_exp=8, track=None, out_f32=0, ev0=None, ev1=None):
        return lib.deqsci_conv3x3_c64_split16(x, w, None, y, n, H, W, 1, w_exp, None, in_exp, None, out_exp, track, out_f32, None, ev0, ev1)
    assert s16(None, p16, q16) == -1
    assert s16(p16, p16, None) == -1 and s16(p16, p16, None, out_f32=1, track=p16) == -4      # y may be NULL only in a measuring launch, which has no fp32 form
    assert s16(p16, p16, p16) == -4                                                            # in place
    assert s16(p16, p16, q16, out_f32=2) == -4                                                 # unknown output form
    assert s16(p16, p16, q16, H=2900, W=2900) == -4                                            # 32-bit offsets / OOB sentinel
    assert s16(p16, p16, q16, in_exp=65) == -4 and s16(p16, p16, q16, out_exp=-65) == -4 and s16(p16, p16, q16, w_exp=100) == -4   # exponents beyond +-64
    assert s16(p16 + 4, p16, q16) == -3
    assert s16(p16, p16, q16, n=0) == -2
    assert s16(p16, p16, q16, ev1=p16) == -1                                                   # one event only
    assert lib.deqsci_f32_to_split16(None, q16, 1, 4, 4, None, 8, None) == -1 and lib.deqsci_f32_to_split16(p16, q16, 1, 0, 4, None, 8, None) == -2
    assert lib.deqsci_f32_to_split16(p16, q16, 1, 4, 4, None, 99, None) == -4
    assert lib.deqsci_absmax_f32(None, 1, 4, p16, None) == -1 and lib.deqsci_absmax_f32(p16, 1, 0, p16, None) == -2 and lib.deqsci_absmax_f32(p16 + 4, 1, 4, p16, None) == -3
    assert lib.deqsci_absmax_f32(p16, 0, 4, p16, None) == -2 and lib.deqsci_absmax_f32(p16, 70000, 4, p16, None) == -4
    assert lib.deqsci_ffdnet_tail_split16(None, p16, q16, 1, 4, 4, 0, None, 8, None) == -1 and lib.deqsci_ffdnet_tail_split16(p16, p16, q16, 1, 4, -4, 0, None, 8, None) == -2
    assert lib.deqsci_conv3x3_c64_to_1_split16(p16 + 4, p16, q16, 1, 4, 4, 0, None, 8, None) == -3
    assert lib.deqsci_ffdnet_head_split16(None, p16, p16, 0, q16, 1, 4, 4, 0, None, 8, None, 8, None, None) == -1
    assert lib.deqsci_ffdnet_head_split16(p16, p16, p16, 0, q16, 1, 4, 4, 0, None, 8, None, 70, None, None) == -4
    assert lib.deqsci_conv3x3_c1_to_64_sp16(None, p16, q16, 1, 4, 4, 1, None, 8, None, None) == -1


def test_act_exp_mirror_matches_the_rule():
    """_hip.act_exp is the host mirror of csrc/common.hpp: sp16_act_exp (the kernels derive it on the device): 2^e amax in [2^11, 2^12),
    zero / subnormal / non-finite -> the default 8, clamped to +-64."""
    import math
    from deqsci_amd import _hip
    for amax in (1.0, 0.999, 10.0, 15.99, 16.0, 255.9, 1e-6, 3.3e-11, 2047.0, 2048.0, 5e4, 1e30, 1e-30):
        e = _hip.act_exp(amax)
        if abs(e) < 64:
            assert 2048 <= amax * 2.0 ** e < 4096, (amax, e)
    assert _hip.act_exp(10.0) == 8 == _hip.SP16_DEFAULT_EXP                     # FFDNet's activations: the fixed scale of round 3
    assert _hip.act_exp(0.0) == 8 and _hip.act_exp(float("inf")) == 8 and _hip.act_exp(float("nan")) == 8 and _hip.act_exp(1e-45) == 8
    assert _hip.act_exp(1e30) == -64 and _hip.act_exp(1e-30) == 64
    src = open(os.path.join(ROOT, "deqsci_amd", "csrc", "common.hpp")).read()
    hdr = open(os.path.join(ROOT, "include", "deqsci_hip.h")).read()
    assert "SP16_DEFAULT_EXP = 8, SP16_TARGET_EXP = 11, SP16_EXP_LIMIT = 64" in src
    assert "#define DEQSCI_SP16_DEFAULT_EXP 8" in hdr and "#define DEQSCI_SP16_TARGET_EXP 11" in hdr
    assert math.isclose(_hip.SP16_ACT_SCALE, 256.0)


def test_shipped_library_reads_no_environment():
    """include/deqsci_hip.h / SURVEY 8(b): "re-entrant, no global state".  The diagnostic knobs (DEQSCI_GRAM_NOISE, DEQSCI_K4_BLOCKS,
    DEQSCI_FORCE_POLICY, DEQSCI_HEAD_VALU) exist only in the -DDEQSCI_DIAG build of `make diag`; the shipped library neither
    names an environment variable nor imports getenv."""
    import subprocess
    from deqsci_amd import _hip
    blob = open(_hip.lib_path(), "rb").read()
    for name in (b"DEQSCI_GRAM_NOISE", b"DEQSCI_K4_BLOCKS", b"DEQSCI_FORCE_POLICY", b"DEQSCI_HEAD_VALU", b"DEQSCI_CONV64"):
        assert name not in blob, name
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _hip.lib_path()], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    csrc = sorted(f for f in os.listdir(os.path.join(ROOT, "deqsci_amd", "csrc")) if f.endswith((".hip", ".hpp")))
    assert "conv_s16.hip" in csrc and "winograd44.hip" in csrc and len(csrc) >= 8
    for src in csrc:
        text = open(os.path.join(ROOT, "deqsci_amd", "csrc", src)).read()
        outside = "".join(seg.split("#endif", 1)[-1] if i else seg for i, seg in enumerate(text.split("#ifdef DEQSCI_DIAG")))
        assert "getenv" not in outside and "static const" not in outside.replace("static constexpr", ""), src


def test_gfx950_code_object_present():
    from deqsci_amd import _hip
    blob = open(_hip.lib_path(), "rb").read()
    assert b"gfx950" in blob and b"gfx90a" not in blob and b"sm_" not in blob


def test_binding_argument_counts_match_header():
    """Every prototype in include/deqsci_hip.h has as many parameters as the ctypes signature the Python
    binding declares for it (a drifted binding would corrupt the call frame silently)."""
    from deqsci_amd import _hip
    src = open(os.path.join(ROOT, "include", "deqsci_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = dict(re.findall(r"\bint\s+(deqsci_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S))
    assert set(_hip.SIGNATURES) <= set(protos)
    for name, argtypes in _hip.SIGNATURES.items():
        params = [p for p in protos[name].split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(argtypes), (name, len(params), len(argtypes))


def _compile_with_resource_report(src_name, tmp_path):
    import subprocess
    src = os.path.join(ROOT, "deqsci_amd", "csrc", src_name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "deqsci_amd", "csrc"), "-c", src, "-o", str(tmp_path / "k.o"), "-Rpass-analysis=kernel-resource-usage",
           "-save-temps=obj"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    asm = [f for f in os.listdir(tmp_path) if f.endswith(".s") and "gfx950" in f]
    assert asm, os.listdir(tmp_path)
    return out.stderr, open(tmp_path / asm[0]).read()


_RESOURCES = r"Function Name: (\S*%s\S*).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)"


def test_winograd44_kernel_has_no_spills(tmp_path):
    """csrc/winograd44.hip sits at the 256-register limit, and ONE spilled register costs it a scratch access plus the
    s_waitcnt vmcnt(0) in front of its use (a memory round trip per stage that also drains the DMA and the stores the design keeps in
    flight); between its asm MFMAs a spill reload would also miss the wait states the compiler gives real MFMAs (VERDICT r2 #2,
    ADVICE r2).  So: every instantiation must compile with zero VGPR / SGPR spills and no scratch, and the code object must not
    contain a single scratch_ instruction.  (Cross-compiles without a GPU, ~10 s.)"""
    report, text = _compile_with_resource_report("winograd44.hip", tmp_path)
    kernels = re.findall(_RESOURCES % "winograd44_conv64_kernel", report, flags=re.S)
    assert len(kernels) == 4, report[-2000:]                           # <IN_BLK, OUT_BLK> in {0,1}^2
    for name, vgprs, scratch, sspill, vspill in kernels:
        assert int(vgprs) <= 256 and (int(scratch), int(sspill), int(vspill)) == (0, 0, 0), (name, vgprs, scratch, sspill, vspill)
    assert text.count("v_mfma_f32_16x16x4_f32") > 1000                  # this is the kernel's code
    assert "scratch_" not in text and "v_writelane" not in text


def test_split16_kernels_have_no_spills(tmp_path):
    """The same guard for csrc/conv_s16.hip: the 64->64 kernel (91 % of a step) runs two waves per SIMD at 255 of 256 registers; a spill
    reload inside a stage is a scratch load plus `s_waitcnt vmcnt(0)`, i.e. a wait for every LDS-DMA instruction in flight (seen while
    moving the tile bookkeeping into the MFMA stream: 2-8 spilled registers, tile_done reloading in the middle of the last stage)."""
    report, text = _compile_with_resource_report("conv_s16.hip", tmp_path)
    kernels = re.findall(_RESOURCES % "(?:conv_s16_kernel|tail_s16_kernel|head_s16_kernel)", report, flags=re.S)
    # conv <0,0,0>, <1,0,0>, <0,1,0> (measuring), <0,0,1> (a run of layers); tail <4,0>, <4,1> (p32 in), <1,0>, <1,1> (SimpleCNN's, p32 in); head <0,0>, <1,0> (measuring), <0,1>, <1,1> (p32 out)
    assert len(kernels) == 12, report[-2000:]
    for name, vgprs, scratch, sspill, vspill in kernels:
        assert int(vgprs) <= 256 and (int(scratch), int(sspill), int(vspill)) == (0, 0, 0), (name, vgprs, scratch, sspill, vspill)
    assert text.count("v_mfma_f32_32x32x16_f16") > 800
    assert "scratch_" not in text and "v_writelane" not in text


def test_wino16_kernels_have_no_spills(tmp_path):
    """The same guard for csrc/conv_w16.hip ("zero-spill guard extended"): the split-fp16 Winograd F(2,3) x direct kernel runs two
    waves per SIMD on 128 accumulators + a ring of two transformed halo rows + eight weight fragments + the tile's deferred outputs; a
    spill inside a half-stage is a scratch access and a vmcnt wait behind every LDS-DMA instruction in flight.  Both instantiations
    (single layer, stack launch) within 256 registers, no scratch; and the stream is what the design says: 8 half-stages x 36 MFMAs per
    instantiation, the transform's subtractions as single v_sub_f32 (hipcc pairs them into v_pk_add_f32 when left alone: 19 cycles beside
    an MFMA, profiles/r05_mfma_f16_fillers.jsonl)."""
    report, text = _compile_with_resource_report("conv_w16.hip", tmp_path)
    kernels = re.findall(_RESOURCES % "conv_w16_kernel", report, flags=re.S)
    assert len(kernels) == 2, report[-2000:]
    for name, vgprs, scratch, sspill, vspill in kernels:
        # (the stack launch's tile bookkeeping holds more scalars than the 102 scalar registers: a dozen of them live in lanes of a vector
        # register between tiles - v_writelane / v_readlane at tile boundaries, never memory)
        assert int(vgprs) <= 256 and (int(scratch), int(vspill)) == (0, 0) and int(sspill) <= (16 if "ILi1E" in name else 0), (name, vgprs, scratch, sspill, vspill)
    assert text.count("v_mfma_f32_32x32x16_f16") == 2 * 8 * 36
    assert "scratch_" not in text
    # packed fp32 additions only where they are written (the epilogue's inline asm, round 6: nothing runs beside it) - never by the compiler
    # in the transform, where one costs 19 cycles beside an MFMA against 2 x 5
    lines = text.splitlines()
    pk = [i for i, ln in enumerate(lines) if "v_pk_add_f32" in ln]
    assert len(pk) == 2 * 64 and all(lines[i - 1].strip() == ";;#ASMSTART" for i in pk), len(pk)

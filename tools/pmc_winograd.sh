#!/bin/bash
# Hardware counters of the two Winograd conv kernels on 64 images of 64 x 128 x 128 (the FFDNet layer shape of the bench workload):
# SQ counters, cache counters, and HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes, no trace domains; corrected as
# tools/pmc_summarize.py calibrates them in the same image: FETCH_SIZE x 2, WRITE_SIZE x 1, both in KiB).
mkdir -p gpurun_out/pmc_wg
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_wg/p$i -o k -- python3 $R/tools/conv_bench.py --shape 64 128 128 > $R/gpurun_out/pmc_wg/p$i.log 2>&1
done
# the stack launch (13 layers over a slice of 32 images in one launch of conv_s16_kernel<0, 0, 1>): HBM bytes per launch
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_wg/stack_$C -o k -- python3 $R/tools/stack_bench.py > $R/gpurun_out/pmc_wg/stack_$C.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json
vals = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in sorted(glob.glob(f'gpurun_out/pmc_wg/stack_{C}/**/k_counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            if "conv_s16_kernel<0, 0, 1>" in r["Kernel_Name"] and r["Counter_Name"] == C:
                v.append(float(r["Counter_Value"]))
    vals[C] = sorted(v)[len(v) // 2] if v else 0.0
n, L, H, W = 32, 13, 128, 128
rd, wr = vals["FETCH_SIZE"] * 1024 * 2.0, vals["WRITE_SIZE"] * 1024 * 1.0          # (units and corrections as tools/pmc_summarize.py calibrates them)
alg = L * (2 * n * H * W * 256 + 64 * 64 * 9 * 4)
json.dump({"kernel": "deqsci::s16::conv_s16_kernel<0, 0, 1> (stack launch: 13 layers over a slice of 32 images)", "shape": [n, 64, H, W], "layers": L,
           "counters_median": vals, "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "hbm_bytes_per_launch": int(rd + wr),
           "algorithmic_hbm_bytes": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 3),
           "note": "algorithmic = every layer reads its input and writes its output once; FETCH_SIZE / WRITE_SIZE count at the L2's fabric side, "
                   "in front of the Infinity Cache: what that cache serves of the slice's activations is not subtracted"}, open("gpurun_out/pmc_conv_s16_stack.json", "w"), indent=1)
PY
python - <<'PY'
import csv, collections, glob, json
n, H, W = 64, 128, 128
for kern, fname, pos, outs, name in (("winograd_conv64_kernel", "gpurun_out/pmc_winograd.json", 16, 4, "deqsci::winograd_conv64_kernel"),
                                     ("winograd44_conv64_kernel<1, 1>", "gpurun_out/pmc_winograd44.json", 36, 16, "deqsci::w44::winograd44_conv64_kernel<1,1> (blk32 -> blk32)"),
                                     ("conv_s16_kernel<0, 0, 0>", "gpurun_out/pmc_conv_s16.json", 0, 0, "deqsci::s16::conv_s16_kernel<0, 0, 0> (sp16 -> sp16)")):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob('gpurun_out/pmc_wg/p*/k_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    med = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
    s16 = pos == 0
    alg = 2 * n * H * W * 64 * 4 + (64 * 64 * 9 * 4 if s16 else pos * 64 * 64 * 4)   # read x once, write y once (256 B per pixel each way) + the weights
    rd, wr = med.get("FETCH_SIZE", 0) * 1024 * 2.0, med.get("WRITE_SIZE", 0) * 1024 * 1.0
    # MFMA instructions: split-fp16 3 x 9 taps x 64 x 64 x 2 flops per pixel on v_mfma_f32_32x32x16_f16 (32768 flops each); Winograd on
    # v_mfma_f32_16x16x4_f32 (2048 flops each)
    mfma = n * H * W * 3 * 9 * 64 * 64 * 2 / 32768 if s16 else n * H * W / outs * pos * 64 * 64 * 2 / 2048
    out = {"kernel": name, "shape": [n, 64, H, W], "launches": len(agg.get("FETCH_SIZE", [])), "counters_median": med,
           "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "hbm_bytes_per_launch": int(rd + wr),
           "algorithmic_hbm_bytes": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 3),
           "mfma_instructions": int(mfma), "non_mfma_valu_per_mfma": round((med.get("SQ_INSTS_VALU", 0) - mfma) / mfma, 3),
           "mfma_busy_fraction": round(med.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * med.get("GRBM_GUI_ACTIVE", 1) / 8), 3),
           "lds_bank_conflict_share": round(med.get("SQ_LDS_BANK_CONFLICT", 0) / max(med.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3)}
    json.dump(out, open(fname, "w"), indent=1)
PY
cat gpurun_out/pmc_winograd.json gpurun_out/pmc_winograd44.json gpurun_out/pmc_conv_s16.json gpurun_out/pmc_conv_s16_stack.json

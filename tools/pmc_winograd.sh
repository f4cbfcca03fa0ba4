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
cd $R
python - <<'PY'
import csv, collections, glob, json
n, H, W = 64, 128, 128
for kern, fname, pos, outs, name in (("winograd_conv64_kernel", "gpurun_out/pmc_winograd.json", 16, 4, "deqsci::winograd_conv64_kernel"),
                                     ("winograd44_conv64_kernel", "gpurun_out/pmc_winograd44.json", 36, 16, "deqsci::w44::winograd44_conv64_kernel")):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob('gpurun_out/pmc_wg/p*/k_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    med = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
    alg = 2 * n * H * W * 64 * 4 + pos * 64 * 64 * 4
    rd, wr = med.get("FETCH_SIZE", 0) * 1024 * 2.0, med.get("WRITE_SIZE", 0) * 1024 * 1.0
    mfma = n * H * W / outs * pos * 64 * 64 * 2 / 2048          # v_mfma_f32_16x16x4_f32: 2048 flops each
    out = {"kernel": name, "shape": [n, 64, H, W], "launches": len(agg.get("FETCH_SIZE", [])), "counters_median": med,
           "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "hbm_bytes_per_launch": int(rd + wr),
           "algorithmic_hbm_bytes": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 3),
           "mfma_instructions": int(mfma), "non_mfma_valu_per_mfma": round((med.get("SQ_INSTS_VALU", 0) - mfma) / mfma, 3),
           "mfma_busy_fraction": round(med.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * med.get("GRBM_GUI_ACTIVE", 1) / 8), 3)}
    json.dump(out, open(fname, "w"), indent=1)
PY
cat gpurun_out/pmc_winograd.json gpurun_out/pmc_winograd44.json

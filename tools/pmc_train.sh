#!/bin/bash
# SQ counters of the update-side convolution kernels (tools/train_kernels_bench.py): where k_trunk_bwd's cycles go
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_train
rocprofv3 -L > gpurun_out/pmc_train/counters.txt 2>&1 || true
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1-2 | tr ' ' '_')
  rm -rf /tmp/pmc_tr
  rocprofv3 --pmc $set --kernel-include-regex "${1:-k_trunk_bwd}" --output-format csv -d /tmp/pmc_tr -o p -- python3 tools/train_kernels_bench.py > gpurun_out/pmc_train/run_$tag.log 2> gpurun_out/pmc_train/err_$tag.log || { tail -5 gpurun_out/pmc_train/err_$tag.log; continue; }
  f=$(find /tmp/pmc_tr -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
    agg[key] += float(r["Counter_Value"]); n[key] += 1
for k, v in sorted(agg.items()): print(k[0], k[1], v / n[k], "per launch over", n[k])
PY
done

#!/bin/bash
# round 5, first probe: what a SIMD sustains (tools/hiptests/t_simd_rate.hip), what the SQ counters read on those known streams, the instruction cache under k_macro_step
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_probe1; mkdir -p $O
tools/hiptests/bin/t_simd_rate > $O/simd_rate.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d /tmp/pmc_simd -o p -- tools/hiptests/bin/t_simd_rate > $O/simd_rate_pmc_stdout.txt 2> $O/simd_rate_pmc_err.log
f=$(find /tmp/pmc_simd -name '*counter_collection.csv' | head -1)
cp "$f" $O/simd_rate_counters.csv
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH --kernel-include-regex 'k_macro_step' --output-format csv -d /tmp/pmc_ic -o p -- python3 tools/physics_rate.py - acorn 144 3000 300 2048 > $O/icache_probe.json 2> $O/icache_err.log
f=$(find /tmp/pmc_ic -name '*counter_collection.csv' | head -1)
python3 - "$f" > $O/icache_counters.txt <<'PY'
import csv, sys, collections
agg = collections.Counter(); n = set()
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print("dispatches", len(n))
for k, v in sorted(agg.items()): print(k, v)
PY
python3 tools/physics_rate.py - acorn 144 3000 1500 2048 > $O/physics_rate_base.json 2> $O/physics_rate_err.log
cat $O/simd_rate.txt $O/icache_counters.txt $O/physics_rate_base.json

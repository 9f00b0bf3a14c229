#!/bin/bash
# instruction-cache and scalar-data-cache counters of the physics-only probe (is the 95 KB kernel thrashing the 64 KB instruction cache?)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_icache
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH --kernel-include-regex 'k_macro_step' --output-format csv -d /tmp/pmc_ic -o p -- python3 tools/physics_rate.py ${1:--} acorn 144 3000 300 2048 > gpurun_out/pmc_icache/probe.json 2> gpurun_out/pmc_icache/err.log
f=$(find /tmp/pmc_ic -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()): print(k, v)
PY

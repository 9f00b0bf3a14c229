#!/bin/bash
# SQ pass of the physics-only probe (tools/physics_rate.py, no render / policy / update): instructions and wave cycles per physics.step().
#   usage: tools/pmc_probe.sh <tag>     -> gpurun_out/pmc_<tag>/probe_SQ.csv + probe.json (merged into the summary by tools/pmc_summary2.py)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$tag
python3 -c "from mujoco_rl_manipulate_unknown_objects_amd import engine; print(engine.source_fingerprint())" > gpurun_out/pmc_$tag/probe_csrc_sha16.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-include-regex 'k_macro_step' --output-format csv -d /tmp/pmc_${tag}_probe -o p -- python3 tools/physics_rate.py - acorn 144 3000 1500 2048 > gpurun_out/pmc_$tag/probe.json 2> gpurun_out/pmc_$tag/probe_err.log
f=$(find /tmp/pmc_${tag}_probe -name '*counter_collection.csv' | head -1)
python3 - "$f" gpurun_out/pmc_$tag/probe_SQ.csv <<'PY'
import csv, sys, collections
agg = collections.Counter(); disp = set()
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["counter", "sum", "dispatches"])
for n, v in agg.items():
    w.writerow([n, v, len(disp)])
PY
cat gpurun_out/pmc_$tag/probe.json

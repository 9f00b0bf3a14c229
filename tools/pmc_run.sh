#!/bin/bash
# Three rocprofv3 --pmc passes of the bench command (SQ counters | FETCH_SIZE | WRITE_SIZE), counters only (gpurun refuses --pmc together
# with the trace domains), into /tmp; only the small per-dispatch CSVs are copied back.  usage: tools/pmc_run.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$tag
python3 -c "from mujoco_rl_manipulate_unknown_objects_amd import engine; print(engine.source_fingerprint())" > gpurun_out/pmc_$tag/csrc_sha16.txt
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $ctrs | cut -d' ' -f1)
  rocprofv3 --pmc $ctrs --kernel-include-regex 'k_macro_step|k_observe|k_conv1_u8' --output-format csv -d /tmp/pmc_${tag}_$name -o p -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 20 "$@" > gpurun_out/pmc_$tag/bench_$name.json 2> gpurun_out/pmc_$tag/err_$name.log
  f=$(find /tmp/pmc_${tag}_$name -name '*counter_collection.csv' | head -1)
  python3 - "$f" gpurun_out/pmc_$tag/$name.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(collections.Counter); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "counter", "sum", "dispatches"])
for k, c in agg.items():
    for n, v in c.items():
        w.writerow([k, n, v, len(disp[k])])
PY
done
ls -la gpurun_out/pmc_$tag

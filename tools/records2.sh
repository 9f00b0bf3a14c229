# second half of the round's final records (bench configurations of BASELINE.json on one GPU, kernel statistics, update-path measurements, a training probe)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
python bench.py --mixed --no-cpu-baseline > gpurun_out/final/mixed.json 2>/dev/null
python bench.py --object bread_crumb --no-cpu-baseline > gpurun_out/final/bread_crumb.json 2>/dev/null
python bench.py --envs 16384 --object sugar_cube --no-cpu-baseline > gpurun_out/final/sugar16384_f32.json 2>/dev/null
python bench.py --envs 16384 --object sugar_cube --state-dtype f16 --no-cpu-baseline > gpurun_out/final/sugar16384_f16.json 2>/dev/null
echo bench-configs-done
rm -rf /tmp/kst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 20 > gpurun_out/final/bench_under_rocprof_steps40.json 2> gpurun_out/final/rocprof.err
cp $(find /tmp/kst -name '*kernel_stats.csv' | head -1) gpurun_out/final/bench_kernel_stats_steps40.csv
echo rocprof-done
{ python tools/update_time.py find 2>&1 | tail -1; python tools/update_kernels.py 2>&1 | grep -v "^\[W\|Warn\|_warn"; python tools/train_kernels_bench.py 2>&1 | tail -4; python tools/policy_fwd_bench.py 2>&1 | tail -14; } > gpurun_out/final/update_path.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/hiptests/t_mfma_peak.hip -o /tmp/t_mfma_peak && /tmp/t_mfma_peak > gpurun_out/final/mfma_issue_microbench.txt
python tools/train_probe.py acorn 75 > gpurun_out/final/train_probe_acorn.log 2>&1
tail -3 gpurun_out/final/train_probe_acorn.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["value"]), d["ms_per_step"])
    except Exception as e: print(f, e)
PY

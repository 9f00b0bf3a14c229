# Round-5 records of the current code on one MI355X: usage tools/records_r5.sh <tag> [quick]
#   bench lines (default = what the driver times at its own flags too), kernel statistics of the bench command (rocprofv3 --kernel-trace --stats),
#   the three --pmc passes of the bench command + the SQ pass of the physics-only probe, phase shares / per-step histograms of the physics kernel.
# Everything lands under gpurun_out/<tag>/ ; the summaries that are kept go to profiles/ by hand (see profiles/README.md).
set -e
tag=$1; quick=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python bench.py > $out/default.json 2> $out/default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/steps20_warmup5.json 2>/dev/null
python tools/physics_rate.py - acorn 144 3000 1500 2048 > $out/physics_rate.json 2>/dev/null
echo bench-done
if [ -z "$quick" ]; then
  python bench.py --actions policy --no-cpu-baseline > $out/policy_actions.json 2>/dev/null
  python bench.py --lockstep --steps 40 --warmup 8 --no-cpu-baseline > $out/lockstep.json 2>/dev/null
  python bench.py --overlap-update --no-cpu-baseline > $out/overlap_update.json 2>/dev/null
  python bench.py --mixed --no-cpu-baseline > $out/mixed.json 2>/dev/null
  python bench.py --object bread_crumb --no-cpu-baseline > $out/bread_crumb.json 2>/dev/null
  python bench.py --envs 16384 --object sugar_cube --no-cpu-baseline > $out/sugar16384_f32.json 2>/dev/null
  python bench.py --envs 16384 --object sugar_cube --state-dtype f16 --no-cpu-baseline > $out/sugar16384_f16.json 2>/dev/null
  echo configs-done
fi
rm -rf /tmp/kst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 20 > $out/bench_under_rocprof_steps40.json 2> $out/rocprof.err
cp $(find /tmp/kst -name '*kernel_stats.csv' | head -1) $out/bench_kernel_stats_steps40.csv
echo rocprof-done
bash tools/pmc_run.sh $tag > $out/pmc_run.log 2>&1 || echo pmc_run-failed
bash tools/pmc_probe.sh $tag > $out/pmc_probe.log 2>&1 || echo pmc_probe-failed
echo pmc-done
if [ -f mujoco_rl_manipulate_unknown_objects_amd/csrc/libgrip_sim_stamps.so ]; then python tools/stamp_async.py acorn 3000 > $out/phase_shares.txt 2>&1 || true; fi
if [ -f mujoco_rl_manipulate_unknown_objects_amd/csrc/libgrip_sim_hist.so ]; then GRIP_STAMPS_LIB=hist python tools/stamp_async.py acorn 3000 > $out/phase_hist.txt 2>&1 || true; fi
python tools/update_time.py find > $out/update_time.txt 2>&1 || true
GRIP_WGRAD23_LIBRARY=1 GRIP_CONV23_F32=1 GRIP_TRUNK_F32=1 python tools/update_time.py find > $out/update_time_round4_kernels.txt 2>&1 || true
python tools/conv23_ab.py > $out/conv23_ab.txt 2>&1 || true
python tools/wgrad23_ab.py > $out/wgrad23_ab.txt 2>&1 || true
if [ -f mujoco_rl_manipulate_unknown_objects_amd/csrc/libgrip_sim_cbst.so ]; then python tools/conv23_stamps.py > $out/conv23_phases.txt 2>&1 || true; fi
if [ -f mujoco_rl_manipulate_unknown_objects_amd/csrc/libgrip_sim_wgst.so ]; then python tools/wgrad23_stamps.py > $out/wgrad23_phases.txt 2>&1 || true; fi
if [ -f mujoco_rl_manipulate_unknown_objects_amd/csrc/libgrip_sim_tbst.so ]; then python tools/trunk_stamps.py > $out/trunk_bwd_phases.txt 2>&1 || true; fi
python tools/update_kernels.py 512 > $out/update_kernels.txt 2>&1 || true
python tools/render_ab.py rays - > $out/render_ab_acorn.txt 2>&1 || true
if [ -f mujoco_rl_manipulate_unknown_objects_amd/csrc/libgrip_sim_capdump.so ]; then
  python tools/newton_cap_probe.py acorn 3000 $out/newton_cap_acorn.json > $out/newton_cap_acorn.txt 2>&1 || true
  python tools/newton_cap_probe.py sugar_cube 3000 $out/newton_cap_sugar_cube.json > $out/newton_cap_sugar_cube.txt 2>&1 || true
fi
echo extras-done
python - "$out" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d.get("value", d.get("env_steps_per_s", 0))), d.get("ms_per_step", d.get("substeps_per_s")))
    except Exception as e: print(f, e)
PY

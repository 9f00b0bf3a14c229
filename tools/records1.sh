set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/default.json 2> gpurun_out/final/default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final/steps20_warmup5.json 2>/dev/null
python bench.py --actions policy --no-cpu-baseline > gpurun_out/final/policy_actions.json 2>/dev/null
python bench.py --lockstep --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/final/lockstep.json 2>/dev/null
python bench.py --overlap-update --no-cpu-baseline > gpurun_out/final/overlap_update.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/*.json")):
    d = json.load(open(f)); print(f, round(d["value"]), d["ms_per_step"])
PY

#!/bin/bash
# same-box A/B of physics-kernel variants (csrc/libgrip_sim_<name>.so, built by tools/build_variant.py; "-" = the shipped library): the physics-only rate of
# tools/physics_rate.py, every variant twice, interleaved.   usage (on the GPU box): tools/ab_physics.sh <out-dir> <name> [<name> ...]
set -e
cd "$GRAFT_REPO_ROOT"; O=$1; shift; mkdir -p $O
for rep in 1 2; do
  for v in "$@"; do
    python3 tools/physics_rate.py $v acorn 144 3000 1500 2048 2>/dev/null | tail -1 > $O/rate_${v/-/base}_$rep.json
    python3 - $O/rate_${v/-/base}_$rep.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:>14s}: {d['substeps_per_s'] / 1e6:7.2f} M physics.step()/s  {d['env_steps_per_s'] / 1e3:6.1f} k env-steps/s  slice kernel {d['slice_kernel_ms']:.3f} ms  faults {d['macro_steps_with_fault_bits_1_2_4']}")
PY
  done
done

#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_probe3; mkdir -p $O
python3 tools/physics_rate.py - acorn 144 3000 1500 2048 > $O/physics_rate.json 2> $O/physics_rate.err
python3 tools/newton_cap_probe.py acorn 3000 $O/cap_acorn.json > $O/cap_acorn.txt 2> $O/cap_acorn.err
python3 tools/newton_cap_probe.py sugar_cube 3000 $O/cap_sugar_cube.json > $O/cap_sugar_cube.txt 2> $O/cap_sugar_cube.err
python3 -m pytest tests/test_gpu_contact.py tests/test_gpu_parity.py tests/test_gpu_state_storage.py -x -q -m gpu > $O/tests.log 2>&1 || true
cat $O/physics_rate.json; tail -8 $O/cap_acorn.txt; tail -8 $O/cap_sugar_cube.txt; tail -5 $O/tests.log

#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_probe2; mkdir -p $O
python3 tools/newton_cap_probe.py acorn 3000 $O/cap_acorn.json > $O/cap_acorn.txt 2> $O/cap_acorn.err
python3 tools/newton_cap_probe.py sugar_cube 3000 $O/cap_sugar_cube.json > $O/cap_sugar_cube.txt 2> $O/cap_sugar_cube.err
python3 tools/stamp_async.py acorn 3000 > $O/stamps_acorn.txt 2>&1
GRIP_STAMPS_LIB=hist python3 tools/stamp_async.py acorn 3000 > $O/hist_acorn.txt 2>&1
tail -30 $O/cap_acorn.txt; tail -30 $O/cap_sugar_cube.txt

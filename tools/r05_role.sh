#!/bin/bash
# k_bt2_role (SPRINGCRAFT_BT2_ROLE = 1): correctness on the batched solver tests, then the C3 bench step A/B
set -u
mkdir -p gpurun_out/r05_role
O=gpurun_out/r05_role
SPRINGCRAFT_BT2_ROLE=1 timeout -k 10 300 python -m pytest tests/test_batched_configs_gpu.py -x -q -k "config4 or pair_chase or config3_batched" > $O/test_role.txt 2>&1; echo "rc $?" >> $O/test_role.txt; tail -4 $O/test_role.txt
grep -q "rc 0" $O/test_role.txt || exit 1
if grep -q "Memory access fault" $O/test_role.txt; then exit 1; fi
SPRINGCRAFT_BT2_ROLE=1 timeout -k 10 600 python -m pytest tests/test_two_stage_gpu.py tests/test_ragged_gpu.py -x -q > $O/test_role2.txt 2>&1; echo "rc $?" >> $O/test_role2.txt; tail -3 $O/test_role2.txt
grep -q "rc 0" $O/test_role2.txt || exit 1
bash tools/r05_ab_env.sh SPRINGCRAFT_BT2_ROLE=0 SPRINGCRAFT_BT2_ROLE=1 3

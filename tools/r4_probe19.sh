#!/bin/bash
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_recovery.py "tests/test_gpu_configs.py::test_config3_bayesr_n50k_m500k_draw_for_draw_against_live_oracle" tests/test_gpu_parity.py -x -q > $O/r4_tests19.log 2>&1; echo "tests rc=$?"; tail -3 $O/r4_tests19.log | head -1
HB_DEBUG_ABORT=1 HB_ROLES_MODEL=BayesR timeout 300 python tools/launch_roles.py 8 2 2>&1 | tail -7
timeout 600 python bench.py --no-ab --tertiary "" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'R', d['secondary']['value'], d['secondary']['roofline']['avg_launch_ms'])"

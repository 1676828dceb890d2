#!/bin/bash
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py tests/test_gpu_c_caller.py tests/test_gpu_configs.py -x -q > $O/r4_tests14.log 2>&1; echo "tests rc=$?"; tail -3 $O/r4_tests14.log | head -2
HB_ROLES_MODEL=BayesR timeout 300 python tools/launch_roles.py 8 2 > $O/r4_roles_bayesr4.txt 2>&1; tail -6 $O/r4_roles_bayesr4.txt
HB_DOTQ2_KIND=2 timeout 300 python tools/launch_roles.py 2 3 > $O/r4_roles_mfma3.txt 2>&1; tail -6 $O/r4_roles_mfma3.txt
timeout 900 python bench.py --tertiary BayesRR --no-cpu > $O/r4_bench_7.json 2> $O/r4_bench_7.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_7.json').read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"]["avg_launch_ms"], "mfma", d["mfma_ab"]["value"], d["mfma_ab"]["roofline"]["avg_launch_ms"], "int8", d["int8"]["value"], "R", d["secondary"]["value"], d["secondary"]["roofline"]["avg_launch_ms"], "RR", d["all_move"][0]["value"])
PY

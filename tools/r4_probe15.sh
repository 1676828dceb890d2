#!/bin/bash
O=gpurun_out
export HB_DEBUG_ABORT=1
HB_ROLES_MODEL=BayesR timeout 300 python tools/launch_roles.py 8 2 2>&1 | tail -7
HB_DOTQ2_KIND=2 timeout 300 python tools/launch_roles.py 2 3 2>&1 | tail -7
timeout 300 python tools/launch_roles.py 2 3 2>&1 | tail -7
unset HB_DEBUG_ABORT
timeout 900 python bench.py --tertiary BayesRR --no-cpu > $O/r4_bench_8.json 2> $O/r4_bench_8.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_8.json').read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"]["avg_launch_ms"], "mfma", d["mfma_ab"]["value"], d["mfma_ab"]["roofline"]["avg_launch_ms"], "int8", d["int8"]["value"], d["int8"]["roofline"]["frac"], "R", d["secondary"]["value"], d["secondary"]["roofline"]["avg_launch_ms"], "RR", d["all_move"][0]["value"])
PY

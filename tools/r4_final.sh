#!/bin/bash
# round-4 closing run: the whole GPU suite, the default bench line, a BayesR soak, on the committed build
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r04_gpu_tests_final.txt 2>&1
grep -n "passed\|failed" $O/r04_gpu_tests_final.txt | tail -2
timeout 900 python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print('value', d['value'], 'int8', d['int8']['value'], 'mfma', d['mfma_ab']['value'], 'secondary', d['secondary']['value'], [(t['model'], round(t['value'],1)) for t in d['all_move']])
PY
timeout 600 python tools/soak.py bayesr 5000 > $O/r04_bayesr_soak.txt 2>&1; tail -3 $O/r04_bayesr_soak.txt
timeout 900 python tools/soak.py dense l 3000 > $O/r04_dense_soak_l.txt 2>&1; tail -2 $O/r04_dense_soak_l.txt

#!/bin/bash
# k_dotq's tile count (HB_DOTQ_TILES, default 768): the int8 headline leg and the dense models
B='python bench.py --steps 60 --warmup 10 --no-cpu --secondary "" --tertiary "" --no-ab --stamped 0 --bits 8'
run() { label="$1"; shift; v=$(env "$@" timeout 120 bash -c "$B" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f sweeps/s, launch %.2f us, frac %.3f' % (d['value'], d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"); echo "BayesCpi int8 | $label: $v"; }
for T in 768 840 900 950 1000; do run "tiles $T" HB_DOTQ_TILES=$T; done
for T in 768 900 1000; do echo "BayesRR | tiles $T: $(HB_DOTQ_TILES=$T timeout 200 python tools/dense_probe.py 2,2 500000 BayesRR 2>&1 | grep -E 'ms per sweep')"; done
B2='python bench.py --steps 40 --warmup 5 --no-cpu --secondary "" --tertiary "" --no-ab --stamped 0 --model BayesR --burnin 300 --bits 8'
for T in 768 900; do echo "BayesR cold int8 | tiles $T: $(HB_DOTQ_TILES=$T timeout 150 bash -c "$B2" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")"; done

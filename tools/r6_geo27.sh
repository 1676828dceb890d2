#!/bin/bash
# round 6, closing: the headline's wide geometry — two or three groups of look-ahead — with the adaptive switch on (the library's behaviour), three runs each,
# and the cold start (sweeps/s over the first 100 / 400 sweeps from bench's regime curve)
B='python bench.py --steps 100 --warmup 10 --no-cpu --secondary "" --tertiary "" --no-ab --stamped 0'
for i in 1 2 3; do for G in 1,3,7 1,2,7; do
  HB_BENCH_GEO_BayesCpi=$G timeout 120 bash -c "$B" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=json.load(open('profiles/bench_last_full.json'))
rc=f.get('regime_curve') or f.get('burnin_curve')
print('geometry $G: %.1f sweeps/s, launch %.2f us; pipeline %s; burn-in curve %s' % (d['value'], d['roofline']['avg_launch_ms']*1e3, d['config']['workload'][-30:], rc))"
done; done

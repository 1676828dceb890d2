#!/bin/bash
# round 6: BayesR (config 3's model) on the group chain — sweeps/s by geometry, 300 sweeps after a cold start and (CONV=1) 2 500 sweeps later
# usage: tools/r6_bayesr_geo.sh "1,3,7 1,2,7 1,2,4 1,2,3 1,2,2 1,2,1"   -> gpurun_out/r06_bayesr_geo.txt
mkdir -p gpurun_out
out=gpurun_out/r06_bayesr_geo${TAG}.txt
: > $out
for geo in $1; do
  for ck in ${CHAINS:-1}; do
    HB_CHAIN=$ck HB_BENCH_GEO_BayesR=$geo python bench.py --steps 20 --warmup 5 --tertiary '' --no-ab --no-cpu --burnin-converged ${CONV:-0} > /tmp/b.json 2> /tmp/b.err
    python - "$geo" "$ck" >> $out <<'P'
import json, sys
full = json.load(open("profiles/bench_last_full.json"))
s = full.get("secondary", {})
c = s.get("converged", {})
def f(b): return "%.1f sweeps/s, %.0f moves/sweep, nnz %s, redo %s, %s, launch %.1f us" % (b["value"], b["mean_changed_markers_per_sweep"], b["NumNZSnp_last"], b.get("chain_rounds_rolled_back_per_sweep"), b.get("regime"), b["roofline"]["avg_launch_ms"] * 1e3) if "value" in b else repr(b)
print("geo %s chain %s bits %s: cold+300 %s | converged %s" % (sys.argv[1], sys.argv[2], s.get("resident_genotype_bits"), f(s), f(c) if c else "-"))
P
    tail -2 /tmp/b.err >> $out
  done
done
cat $out

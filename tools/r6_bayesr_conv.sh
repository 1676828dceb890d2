#!/bin/bash
# round 6: BayesR in the converged regime (continued from profiles/state/bayesr_config3.npz) — sweeps/s per geometry / knob
# usage: tools/r6_bayesr_conv.sh TAG "ENV1=.. ENV2=..|ENV..." "geo geo ..."
mkdir -p gpurun_out
out=gpurun_out/r06_bayesr_conv_$1.txt
: > $out
IFS='|' read -ra ENVS <<< "$2"
for geo in $3; do
  for e in "${ENVS[@]}"; do
    env $e HB_BENCH_GEO_BayesR=$geo python bench.py --steps 40 --warmup 5 --burnin 100 --burnin-secondary 20 --tertiary '' --no-ab --no-cpu > /tmp/b.json 2> /tmp/b.err
    python - "$geo" "$e" >> $out <<'P'
import json, sys
full = json.load(open("profiles/bench_last_full.json"))
c = full.get("secondary", {}).get("converged", {})
print("geo %s [%s] bits %s: converged %s" % (sys.argv[1], sys.argv[2], c.get("resident_genotype_bits"), "%.1f sweeps/s, %.0f moves/sweep, nnz %s, redo %.1f, %s, launch %.1f us, stamped sweep %.3f ms" % (c["value"], c["mean_changed_markers_per_sweep"], c["NumNZSnp_last"], c.get("chain_rounds_rolled_back_per_sweep") or 0, c.get("regime"), c["roofline"]["avg_launch_ms"] * 1e3, c["roofline"]["in_situ"]["ms_per_step_of_the_stamped_sweeps"]) if "value" in c else repr(c)))
P
  done
done
cat $out

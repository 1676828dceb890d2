#!/bin/bash
# round 6, closing check: BayesR converged (from the stored state, group chain at (2, 2)) and 300 sweeps after a cold start, under the tuning knobs
R=${GRAFT_REPO_ROOT:-.}
BC="python bench.py --steps 60 --warmup 10 --no-cpu --secondary \"\" --tertiary \"\" --no-ab --stamped 0 --model BayesR --init-state $R/profiles/state/bayesr_config3.npz --burnin 40"
BK="python bench.py --steps 40 --warmup 5 --no-cpu --secondary \"\" --tertiary \"\" --no-ab --stamped 0 --model BayesR --burnin 300"
run() { what="$1"; shift; label="$1"; shift; if [ "$what" = conv ]; then B="$BC"; else B="$BK"; fi
  v=$(env "$@" timeout 150 bash -c "$B" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f sweeps/s (%s)' % (d['value'], d['config']['workload'][-28:]))"); echo "$what | $label: $v"; }
run conv "defaults" HB_X=0
run conv "defaults again" HB_X=0
run conv "HB_WARM_G=0" HB_WARM_G=0
run conv "HB_WARM_G=2" HB_WARM_G=2
run conv "HB_WARM_G=8" HB_WARM_G=8
run conv "HB_CANDF=0.6" HB_CANDF=0.6
run conv "HB_CANDF=0.9" HB_CANDF=0.9
run conv "HB_CANDF=1.0" HB_CANDF=1.0
run conv "HB_KAPPA=2" HB_KAPPA=2
run conv "HB_KAPPA=6" HB_KAPPA=6
run conv "HB_FWD2_OFF=1" HB_FWD2_OFF=1
run conv "HB_WARM_AHEAD=2" HB_WARM_AHEAD=2
run conv "HB_WARM_AHEAD=8" HB_WARM_AHEAD=8
run cold "defaults" HB_X=0
run cold "HB_WARM_R=0" HB_WARM_R=0
run cold "HB_WARM_R=8" HB_WARM_R=8
run cold "HB_KAPPA=2" HB_KAPPA=2
run cold "HB_KAPPA=6" HB_KAPPA=6
run cold "HB_FWD_R=0" HB_FWD_R=0

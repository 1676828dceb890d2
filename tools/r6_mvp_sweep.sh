#!/bin/bash
# round 6: the persistent mat-vec (HB_MVP=1) under different poll pacings; prints sweeps/s and how many sweeps were replayed after a time-out
B='python bench.py --burnin 20 --steps 20 --warmup 3 --stamped 0 --no-cpu --secondary "" --tertiary "" --no-ab'
run() { # label, env...
  local label="$1"; shift
  out=$(env "$@" HB_MVP=1 HB_NO_ADAPTIVE=1 timeout 90 bash -c "$B" 2>&1)
  v=$(echo "$out" | tail -1 | python -c "import sys,json
try: print(json.loads(sys.stdin.read())['value'])
except Exception as e: print('no line')")
  n=$(echo "$out" | grep -c "timed out")
  echo "$label: $v sweeps/s, time-outs reported $n"
}
run "default pacing"
run "tiles nap 1, fresh 2" HB_MVP_TSLEEP=1 HB_MVP_TFRESH=2
run "tiles nap 2, fresh 1" HB_MVP_TSLEEP=2 HB_MVP_TFRESH=1
run "tiles nap 2 fresh 1, update nap 1 fresh 2" HB_MVP_TSLEEP=2 HB_MVP_TFRESH=1 HB_MVP_USLEEP=1 HB_MVP_UFRESH=2
run "tiles nap 4 fresh 1, update nap 2 fresh 1" HB_MVP_TSLEEP=4 HB_MVP_TFRESH=1 HB_MVP_USLEEP=2 HB_MVP_UFRESH=1
run "same, chain fresh 1" HB_MVP_TSLEEP=4 HB_MVP_TFRESH=1 HB_MVP_USLEEP=2 HB_MVP_UFRESH=1 HB_MVP_CFRESH=1
run "narrow (2,2), default pacing" HB_BENCH_GEO_BayesCpi=1,2,2

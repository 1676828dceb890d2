#!/bin/bash
TUNES="6,0.64 4,0.64 5,0.64 8,0.64 6,0.5 6,0.75" timeout 800 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 40 2>&1 | grep kappa

#!/bin/bash
O=gpurun_out
HB_ROLES_MODEL=BayesR HB_DEBUG_ABORT=1 timeout 300 python tools/launch_roles.py 8 2 > $O/r4_roles_bayesr_final.txt 2>&1; tail -12 $O/r4_roles_bayesr_final.txt | cut -c1-250

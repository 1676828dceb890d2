#!/bin/bash
O=gpurun_out
HB_ROLES_MODEL=BayesR timeout 300 python tools/launch_roles.py 8 2 > $O/r4_roles_bayesr.txt 2>&1; tail -7 $O/r4_roles_bayesr.txt
HB_FWD_R=1 HB_ROLES_MODEL=BayesR timeout 300 python tools/launch_roles.py 8 2 > $O/r4_roles_bayesr_nofwd.txt 2>&1; tail -7 $O/r4_roles_bayesr_nofwd.txt

#!/bin/bash
# tools/kernel_resources.sh [pattern]: registers, spills, scratch and occupancy per kernel of hb_kernels.hip, as the compiler reported them in the last build
# (hibayes_amd/csrc/hb_kernels.res.txt, written by the Makefile). A spill in a chain kernel is a regression: tests/test_host_logic.py checks the headline's.
cd "$(dirname "$0")/.."
make -C hibayes_amd/csrc -j4 >/dev/null || exit 1
python3 tools/kernel_resources.py hibayes_amd/csrc/hb_kernels.res.txt "$1"

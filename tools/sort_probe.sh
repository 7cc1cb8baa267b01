#!/bin/bash
B=fastga_amd/bin/sort_bench
run() { echo "## $*"; timeout 120 env "$@" 2>&1 | tail -1 | cut -c1-150; }
run FGA_SORT_NT=256 FGA_SORT_PH=3 $B 1000003 53 12 uniform 1
run FGA_SORT_NT=512 FGA_SORT_KPT=8 FGA_SORT_PH=3 $B 1000003 53 12 uniform 1
for cfg in "256 16 3" "512 16 4" "512 8 3" "512 8 4" "1024 8 4" "1024 8 5"; do set -- $cfg; run FGA_SORT_NT=$1 FGA_SORT_KPT=$2 FGA_SORT_PH=$3 $B 550000000 61 12 uniform 3; done

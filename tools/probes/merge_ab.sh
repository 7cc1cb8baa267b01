#!/bin/bash
# tools/merge_ab.sh <variant names...> -- merge_bench.py under the default library and each fastga_amd/variants/lib_<name>.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in default "$@"; do
  if [ $v = default ]; then unset FGA_LIBRARY; else export FGA_LIBRARY=$PWD/fastga_amd/variants/lib_$v.so; fi
  echo "== $v"
  timeout 100 python tools/merge_bench.py --reps 4 --check $MB_ARGS 2>&1 | tail -1
done

#!/bin/bash
# tools/profile_local.sh <tag> -- regenerate profiles/<tag>_* from a CLEAN tree at HEAD: refuses when the tree has
# uncommitted changes, passes the commit to the GPU box (gpurun ships no .git), copies the summaries into profiles/.
# (the committed profiles must have been taken at the commit they name.)
set -e
tag=${1:-r03}
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- . ':!profiles' ':!gpurun_out')" ]; then
  echo "profile_local.sh: the tree has uncommitted changes; commit first" >&2
  git status --short | head >&2
  exit 1
fi
commit=$(git rev-parse --short HEAD)
make -s -C fastga_amd/csrc -j8 > /dev/null
gpurun --timeout 2700 -- "bash tools/profile_round.sh $tag $commit > gpurun_out/${tag}_profile_round.log 2>&1; bash tools/scale_prof.sh $tag > gpurun_out/${tag}_scale_prof.log 2>&1; tail -5 gpurun_out/${tag}_profile_round.log"
for f in kernel_stats.csv pmc_summary.csv bench_under_rocprof.json config4_kernel_stats.csv config3_kernel_stats.csv throughput_kernel_stats.csv \
         config4_pmc_summary.csv config3_pmc_summary.csv throughput_pmc_summary.csv; do
  [ -f gpurun_out/${tag}_$f ] && [ gpurun_out/${tag}_$f -nt profiles/${tag}_commit.txt -o ! -f profiles/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
sed -i "1s/.*/# commit: $commit/" profiles/${tag}_pmc_summary.csv
echo "$commit  (tools/profile_local.sh $tag: profile_round.sh + scale_prof.sh on the GPU box)" > profiles/${tag}_commit.txt
echo "profiles/${tag}_* regenerated at $commit"

#!/bin/bash
# usage: pmc_run.sh <tag> <counters...>   (one rocprofv3 pass per counter; SNSDE_LIB / SNSDE_NO_LEAN taken from the environment)
# writes gpurun_out/pmc_<tag>/<counter>/... and prints the per-kernel summary
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/$c
  mkdir -p $out
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o run -- python $GRAFT_REPO_ROOT/tools/lean_check.py time > $out/log.txt 2>&1
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f _kernel | grep -E "^  $c|^void|grid" | head -4
done

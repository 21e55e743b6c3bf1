#!/bin/bash
# usage: pmc_multi.sh "<counters>" -- <run_kernel args> [-- <run_kernel args> ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ctrs="$1"; shift; shift
args=()
run() { echo "== ${args[*]}"; bash tools/pmc_one.sh "$ctrs" "${args[@]}" 2>&1 | awk '{print $1, $2, $NF}' | cut -c1-110; }
for a in "$@"; do
  if [ "$a" == "--" ]; then run; args=(); else args+=("$a"); fi
done
run

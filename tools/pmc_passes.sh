#!/bin/bash
# several rocprofv3 --pmc passes over one script, one counter set per pass, each on a leash (a counter set the hardware cannot
# collect aborts the profiled process and can leave rocprofv3 waiting): tools/pmc_passes.sh "SET A" "SET B" ... -- script.py [args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
for s in "${sets[@]}"; do
  echo "== $s"
  timeout -k 10 ${PMC_PASS_TIMEOUT:-180} bash $ROOT/tools/pmc_cmd.sh "$s" "$@" 2>&1 | grep -v "^W2\|^    @\|^E2" | tail -${PMC_PASS_LINES:-14}
done

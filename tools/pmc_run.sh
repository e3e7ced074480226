#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <counters...> -- python3 script args   (separate pass per call)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$out" -- "$@" > /dev/null 2>&1

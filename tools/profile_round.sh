#!/bin/bash
# Round profile (run ON the GPU box): the current round's script.  tools/profile_r04.sh: bench line, rocprofv3 kernel-trace stats of the
# same command, PMC passes over the bench launch and the aligned phase; then (here, off the box) tools/pmc_bench_summary.py TAG 10000 100000 40 1500
# writes profiles/pmc_summary.json with the sha of the device sources.  Earlier rounds: profile_r02.sh, profile_r03.sh.
exec "$(dirname "$0")/profile_r04.sh" "$@"

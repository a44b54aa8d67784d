#!/bin/bash
# Profile a command on the GPU box with rocprofv3 and leave CSV summaries under gpurun_out/<name>/.
#   tools/prof.sh <name> stats            -- <program> <args...>     kernel trace + per-kernel stats
#   tools/prof.sh <name> pmc "<counters>" -- <program> <args...>     one PMC pass (kernel trace only; never with other traces)
# The program itself must follow "--" (python3 script.py ...): no env/bash -c hops under the profiler.
set -e
name=$1; mode=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$name
mkdir -p "$out"
export TMPDIR=/tmp
export BUF_BENCH_NO_FORK=1          # bench.py: no forked sample generation under the profiler (a forked child can hang in the tool's finaliser)
args=()
for a in "$@"; do                      # the profiler runs from /tmp: make repo-relative paths absolute
    if [ -e "$R/$a" ] && [ "${a#-}" = "$a" ]; then args+=("$R/$a"); else args+=("$a"); fi
done
set -- "${args[@]}"
cd /tmp
if [ "$mode" = stats ]; then
    shift   # --
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$name" -- "$@" > "$out/run.log" 2>&1
else
    counters=$1; shift 2
    timeout 900 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$out" -o "$name" -- "$@" > "$out/run.log" 2>&1
fi
cd "$R"
find "$out" -name "*.csv" | head

#!/bin/bash
# rocprofv3 kernel stats of one command, printed compactly and copied to gpurun_out/r05/<name>_kernel_stats.csv
# usage (on the GPU box): bash experiments/r05_prof.sh <name> <command...>
name=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$name && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- "$@" > $R/gpurun_out/r05/${name}_stdout.txt 2> $R/gpurun_out/r05/${name}_stderr.txt < /dev/null )
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no kernel stats produced"; exit 1; fi
cp "$f" $R/gpurun_out/r05/${name}_kernel_stats.csv; t=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1); [ -n "$TIMELINE" ] && python3 $R/experiments/r05_timeline.py "$t" "$TIMELINE" > $R/gpurun_out/r05/${name}_timeline.txt
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for cut in ("<", "("):
        pass
    short = n.replace("void ", "").replace("wm::(anonymous namespace)::", "").replace("wm::split::", "split::")[:70]
    print("%-70s calls %5s avg %10.1f us  min %9.1f  max %9.1f" % (short, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY

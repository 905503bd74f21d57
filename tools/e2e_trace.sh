#!/bin/bash
# kernel trace of the e2e_stage2 pass (the bench's other passes switched off); summary -> gpurun_out/<tag>/kstats.txt
tag=${1:-e2e_trace}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0 --e2e-depth5-streams 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o s -- python3 bench.py $ARGS > $out/line.json 2> $out/err.txt
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 tools/kstats.py $out/prof | head -40
cp "$f" $out/kernel_stats.csv

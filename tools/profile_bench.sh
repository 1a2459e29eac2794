#!/bin/bash
# GPU box: `rocprofv3 --kernel-trace --stats` over the driver's bench command (configs, prg_mode, c1_additive and the open step
# included), the per-kernel summary into gpurun_out/<tag>_bench_full_kernel_stats.csv and the line into <tag>_bench_profiled.json.
# The program itself follows `--` (no env/bash hop); no --pmc here (tools/regen_pmc.sh does the counters in passes of their own).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r3}
export TMPDIR=/tmp
cd "$R" || exit 1
rm -rf gpurun_out/prof_stats
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats --output-format csv -- \
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/${TAG}_bench_profiled_detail.json > gpurun_out/${TAG}_bench_profiled.json 2> gpurun_out/${TAG}_bench_profiled.err || exit 2
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || exit 3
cp "$f" gpurun_out/${TAG}_bench_full_kernel_stats.csv
rm -rf gpurun_out/prof_stats
head -25 gpurun_out/${TAG}_bench_full_kernel_stats.csv

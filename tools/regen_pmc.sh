#!/bin/bash
# GPU box: the two PMC passes behind profiles/pmc_traffic.json, then the file itself (into gpurun_out/; copy it to profiles/).
# rocprofv3 gets the program itself after `--` (no env/bash hop), and --pmc is never combined with a trace domain.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- \
  python3 bench.py --open 0 --cpu-sample 0 --ew 0 --steps 5 --pmc-live 0 --detail gpurun_out/pmc_fetch_detail.json > gpurun_out/pmc_fetch.json 2> gpurun_out/pmc_fetch.err || exit 2
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- \
  python3 bench.py --open 0 --cpu-sample 0 --ew 0 --steps 5 --pmc-live 0 --detail gpurun_out/pmc_write_detail.json > gpurun_out/pmc_write.json 2> gpurun_out/pmc_write.err || exit 3
python3 tools/make_pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 4000000000 > gpurun_out/pmc_traffic.json || exit 4
# the raw counter CSVs are large; keep only the summary
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
cat gpurun_out/pmc_traffic.json

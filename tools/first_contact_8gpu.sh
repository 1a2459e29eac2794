#!/bin/bash
# First contact with an 8-GPU MI355X node -- everything in this repository that needs more than one real GPU, in the order that
# makes the first run count.  No multi-GPU number in DESIGN.md is a measurement until this has run: the open step's RCCL code
# (csrc/open_rccl.inc: grouped per-row all-gathers, double buffering, reduce-scatter) has only met tests/cxx/fake_rccl.cc and
# one-rank communicators.
#
#   bash tools/first_contact_8gpu.sh [outdir]        (default gpurun_out/first_contact; needs 8 visible GPUs)
#
# Every step is a fresh process started BEFORE anything in it touches a GPU (python -m torch.distributed.run / bench.py's own
# launcher; no exec from a GPU-initialised process), bounded by `timeout -k`, joined with && -- a step that fails or hangs
# stops the script: read its log before running anything again.
#   (a) tools/open_rccl_check.py on 2, 4, 8 ranks: scl_hip_open_all_gather / _partial_gather / _reduce_scatter and their
#       torch.distributed twins against the CPU oracle, 10^5 secrets, three fields
#   (b) bench.py --gpus 1, 2, 4, 8: the headline (BASELINE configs[1], weak scaling) -- the compact line of each into
#       scale_N.json, checked: rccl.ranks == N, N entries in ms_per_step_by_rank, verified
#   (c) bench.py --gpus 8 --config c4 and --config c5: the configurations BASELINE quotes on 8 GPUs, with rccl_busbw_GBps
# The N = 1 line of (b) is the driver's BENCH command's workload (tests/test_bench_launcher.py checks that on a dry run).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/first_contact}
DRY=${FIRST_CONTACT_DRY_RUN:-0}      # 1: gloo + --dry-run through every bench command, step (a) skipped (tests, no GPU)
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
cd "$R" || exit 1
mkdir -p "$OUT" || exit 1
port() { python3 -c 'import socket; s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])'; }
check_line() {   # file, ranks
  python3 - "$1" "$2" <<'PY' || return 1
import json, sys
line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n = int(sys.argv[2])
assert line["n_gpus"] == n and line["rccl"]["ranks"] == n and len(line["ms_per_step_by_rank"]) == n, line.get("rccl")
assert len(json.dumps(line)) < 8192
if line.get("data") != "none (dry run)":
    assert line["verified"] is True, line.get("errors")
    assert n == 1 or line["rccl"].get("allreduce_of_ones") == n, line["rccl"]
print(f"  n_gpus={n}: value={line['value']:.4g} {line['unit']}, ms_per_step={line['ms_per_step']:.4g}, workload: {line['config']['workload']}")
PY
}
EXTRA=()
[ "$DRY" = 1 ] && EXTRA=(--backend gloo --dry-run)

if [ "$DRY" != 1 ]; then
  NGPU=$(python3 -c 'import torch; print(torch.cuda.device_count())')
  [ "$NGPU" -ge 8 ] || { echo "first_contact: $NGPU GPUs visible, 8 needed"; exit 2; }
  for W in 2 4 8; do
    echo "== (a) open step over real RCCL, $W ranks"
    timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port "$(port)" \
      tools/open_rccl_check.py > "$OUT/open_rccl_$W.json" 2> "$OUT/open_rccl_$W.err" || { echo "FAILED: see $OUT/open_rccl_$W.err"; tail -5 "$OUT/open_rccl_$W.err"; exit 3; }
    tail -1 "$OUT/open_rccl_$W.json"
  done
fi
for N in 1 2 4 8; do
  echo "== (b) headline, $N GPU(s)"
  timeout -k 10 900 python3 bench.py --gpus $N --steps 20 --warmup 5 --detail "$OUT/scale_${N}_detail.json" "${EXTRA[@]}" \
    > "$OUT/scale_$N.json" 2> "$OUT/scale_$N.err" || { echo "FAILED: see $OUT/scale_$N.err"; tail -5 "$OUT/scale_$N.err"; exit 4; }
  check_line "$OUT/scale_$N.json" $N || { echo "FAILED: the line of $N rank(s) does not check"; exit 5; }
done
for C in c4 c5; do
  echo "== (c) --config $C on 8 GPUs"
  timeout -k 10 1500 python3 bench.py --gpus 8 --config $C --detail "$OUT/${C}_8_detail.json" "${EXTRA[@]}" \
    > "$OUT/${C}_8.json" 2> "$OUT/${C}_8.err" || { echo "FAILED: see $OUT/${C}_8.err"; tail -5 "$OUT/${C}_8.err"; exit 6; }
  check_line "$OUT/${C}_8.json" 8 || { echo "FAILED: the line of --config $C does not check"; exit 7; }
done
python3 - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
vals = {n: json.loads(open(os.path.join(out, f"scale_{n}.json")).read().strip().splitlines()[-1])["value"] for n in (1, 2, 4, 8)}
if vals[1]:
    print("weak scaling (value / (N x value at 1)):", {n: round(v / (n * vals[1]), 3) for n, v in vals.items()})
PY
echo "first contact complete: $OUT"

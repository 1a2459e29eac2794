#!/bin/bash
# GPU box: SQ counters of C4's two kernels and of k_prg_blocks (tools/run_c4_once.py), rocprofv3 --pmc only, the program itself
# after `--`; prints the mean per launch of every counter for the kernels named below (profiles/r3_c4_sq.txt).
export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INST_CYCLES_SALU"; do
  rm -rf /tmp/sqp
  timeout -k 10 200 rocprofv3 --pmc $pass -d /tmp/sqp --output-format csv -- python3 tools/run_c4_once.py > /dev/null 2>/tmp/sqp.err || { echo "pass failed: $pass"; tail -3 /tmp/sqp.err; continue; }
  python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob("/tmp/sqp/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        for name in ("k_share_gf_tiles", "k_recover_gf128_pos", "k_prg_blocks"):
            if name in k:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    for c, v in sorted(cs.items()):
        v = sorted(v)[-2:] if name == "k_recover_gf128_pos" else v   # (the first-use self-check launches are small)
        print(f"{name:22s} {c:28s} {len(v):3d} {sum(v)/len(v):18.1f}")
PY
done

#!/bin/bash
# GPU box: SQ counters of k_share_mfma_m61_p16 in tools/mfma_bench builds (profiles/r3_p16_sq.txt).  Expects
# tools/_build/mfma_bench_<variant> for the variants below (o1: -DMF16_STAGGER=0, o1a4: + -DMF16_ABL=4, o1s: as shipped).
# --pmc only, the program itself after `--`.
export TMPDIR=/tmp
for b in o1 o1a4 o1s; do
  for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VMEM" ; do
    rm -rf /tmp/sqp; timeout -k 10 120 rocprofv3 --pmc $pass -d /tmp/sqp --output-format csv -- tools/_build/mfma_bench_$b 128 42 20000000 > /dev/null 2>/tmp/sqp.err || { echo "pass failed: $pass"; tail -3 /tmp/sqp.err; continue; }
    echo "== $b"; python3 tools/sq_means.py /tmp/sqp
  done
done

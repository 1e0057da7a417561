#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/pmc_pass.sh <outdir> <program> [args...]
# One rocprofv3 run per counter group (the guide's rule: --pmc passes on their own, no trace domains beside them), then one
# --kernel-trace --stats run.  Fails loudly: every pass's stderr is kept in <outdir>/<pass>.log and a failed pass stops the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun, from the repo root)}"
out=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$out"
export TMPDIR=/tmp
cd /tmp
run_pass() {  # name, counters...
  local name=$1; shift
  echo "[pmc_pass] $name: $*" >&2
  if ! rocprofv3 --pmc "$@" --output-format csv -d "$R/$out/$name" -- python3 "$R/$PROG" "${PARGS[@]}" > "$R/$out/$name.log" 2>&1; then
    echo "[pmc_pass] pass $name FAILED; tail of its log:" >&2; tail -n 20 "$R/$out/$name.log" >&2; exit 1
  fi
}
PROG=$1; shift
PARGS=("$@")
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run_pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
echo "[pmc_pass] kernel trace" >&2
if ! rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/trace" -- python3 "$R/$PROG" "${PARGS[@]}" > "$R/$out/trace.log" 2>&1; then
  echo "[pmc_pass] trace pass FAILED" >&2; tail -n 20 "$R/$out/trace.log" >&2; exit 1
fi
python3 "$R/tools/pmc_table.py" "$R/$out" > "$R/$out/summary.txt"
echo "[pmc_pass] done: $out/summary.txt" >&2

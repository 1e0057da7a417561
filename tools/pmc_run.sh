#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/pmc_run.sh <gemm|attn> <outdir>
# separate --pmc passes (no trace domains), as gpurun requires
what=$1; out=$2; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$out/sq -- python3 $R/tools/gemm_only.py $what > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/$out/sq2 -- python3 $R/tools/gemm_only.py $what > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/$out/tcc -- python3 $R/tools/gemm_only.py $what > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/fetch -- python3 $R/tools/gemm_only.py $what > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/write -- python3 $R/tools/gemm_only.py $what > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python3 $R/tools/gemm_only.py $what > /dev/null 2>&1
find $R/$out -name "*.csv" | head -20

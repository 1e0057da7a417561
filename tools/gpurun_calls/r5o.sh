set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5o
mkdir -p $O
cd $R
for v in "FLEXAM_SP_PIECES=2 FLEXAM_SP_OVERLAP=1" "FLEXAM_SP_PIECES=1 FLEXAM_SP_OVERLAP=1" "FLEXAM_SP_PIECES=1 FLEXAM_SP_OVERLAP=0" "FLEXAM_SP_PIECES=2 FLEXAM_SP_OVERLAP=0" "FLEXAM_SP_PIECES=3 FLEXAM_SP_OVERLAP=1"; do echo "$v"; env $v python tools/emulate_rank.py 8 0 6 2 2>&1 | tail -1; done > $O/emulate8_gather_variants.txt
cat $O/emulate8_gather_variants.txt

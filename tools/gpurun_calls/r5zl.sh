set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zl
mkdir -p $O
cd $R
AB_ROUNDS=9 python tools/ab_attn_variants.py tree nomax > $O/nomax.txt 2>&1; cat $O/nomax.txt

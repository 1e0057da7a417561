# r6f: 30 layers x d = 3072, ONE forward at L = 11648 (configs[1]'s token count, one sample) against the fp32 oracle: seeded weights and logit std 6
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f
mkdir -p $O
cd $R
python tools/parity_30_layers.py --h 32 --w 56 --batch 1 --host-oracle seeded > $O/parity_30_layers_L11648.txt 2>&1; echo "rc $?" >> $O/parity_30_layers_L11648.txt
tail -12 $O/parity_30_layers_L11648.txt

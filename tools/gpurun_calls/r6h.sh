# r6h: kernel trace of the full-size VAE encode (4 encodes: warm-up + 3 timed) with the 384 / 320-row 160-wide tiles
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6h
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/vae_encode_ab.py --child > $O/encode.txt 2>&1
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/vae_encode_kernel_stats.csv; rm -rf $O/trace
tail -1 $O/encode.txt

# r6zb: deferred-rescale threshold sweep (2^0 ... 2^24): attention error against fp64 on peaked rows, and the in-step time at logit std 6
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6zb
mkdir -p $O
cd $R
python tools/attn_rescale_threshold_error.py 2>&1 | grep -v amdgpu.ids > $O/error_vs_threshold.txt
timeout 900 python tools/ab_step.py lib=thr8 lib=thr4 lib=thr12 lib=thr16 lib=tree --logit=6 --steps=3 --rounds=5 2>&1 | grep -v amdgpu.ids > $O/ab_step_logit6.txt
cat $O/error_vs_threshold.txt; tail -6 $O/ab_step_logit6.txt

# r6za: deferred-rescale threshold 2^8 -> 2^24 in the bf16 attention kernel: parity tests (peaked rows included), then the in-step A/B against a
# -DA32_RESCALE_THR=8 build on peaked rows (--logit=6) and on the seeded weights
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6za
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_full_depth_gpu.py tests/test_dit_gpu.py tests/test_sp_gpu.py -q -m gpu -x > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
timeout 600 python tools/ab_step.py lib=thr8 lib=tree --logit=6 --steps=3 --rounds=5 > $O/ab_step_logit6.txt 2>&1; echo "ab logit rc $?" >> $O/rc.txt
timeout 600 python tools/ab_step.py lib=thr8 lib=tree --steps=3 --rounds=5 > $O/ab_step_seeded.txt 2>&1; echo "ab seeded rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/tests.txt; tail -4 $O/ab_step_logit6.txt; tail -4 $O/ab_step_seeded.txt

# r6l: SAGE_ATTENTION under the K|V gather (MXFP8 records gathered, chunked key records in the kernel): SP tests, MXFP8 kernel tests, its timing
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6l
mkdir -p $O
cd $R
python -m pytest tests/test_attn_fp8_gpu.py -q -m gpu > $O/tests_attn_fp8.txt 2>&1; echo "tests_attn_fp8 rc $?" >> $O/rc.txt
python -m pytest tests/test_sp_gpu.py -q -m gpu -s -k "sage" > $O/tests_sp_sage.txt 2>&1; echo "tests_sp_sage rc $?" >> $O/rc.txt
python tools/check_attn_fp8.py > $O/check_attn_fp8.txt 2>&1
python -m pytest tests/test_replay_gpu.py tests/test_full_width_gpu.py -q -m gpu -k "replay or quantised or sage or configs4" > $O/tests_misc.txt 2>&1; echo "tests_misc rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -n 4 $O/tests_attn_fp8.txt $O/tests_sp_sage.txt $O/tests_misc.txt; grep -h "rel-rms" $O/tests_sp_sage.txt | tail -8; tail -6 $O/check_attn_fp8.txt

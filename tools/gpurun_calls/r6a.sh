# r6a: the round's first call -- new / changed GPU tests, the oracle-on-GPU check of tools/parity_30_layers.py, a bench line with the new emulated-rank host figure
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a
mkdir -p $O
cd $R
python -m pytest tests/test_raster_gpu.py tests/test_conv_helpers_gpu.py tests/test_fp8_gpu.py -x -q -m gpu -s > $O/tests_a.txt 2>&1; echo "tests_a rc $?" >> $O/rc.txt
python -m pytest tests/test_sp_gpu.py -q -m gpu -s > $O/tests_sp.txt 2>&1; echo "tests_sp rc $?" >> $O/rc.txt
python -m pytest "tests/test_full_width_gpu.py" -q -m gpu -s -k "configs4" > $O/tests_configs4.txt 2>&1; echo "tests_configs4 rc $?" >> $O/rc.txt
python tools/parity_30_layers.py --layers 2 --host-oracle all > $O/parity_2_layers_L2912.txt 2>&1; echo "parity rc $?" >> $O/rc.txt
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-clip > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/tests_a.txt $O/tests_sp.txt $O/tests_configs4.txt; tail -4 $O/parity_2_layers_L2912.txt

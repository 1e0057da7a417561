# usage (GPU box): bash tools/stress_8ranks.sh [runs]   -- the full-size 8-rank one-device run (gloo) of every pinned layout, several times: the self-check value
# of a layout must repeat bit for bit (the same kernels on the same data); a value that moves is a race.
runs=${1:-5}
for lay in "allgather 1" "ulysses 0" "ulysses 1"; do set -- $lay
  for i in $(seq 1 $runs); do FLEXAM_SP_MODE=$1 FLEXAM_CFG_PARALLEL=$2 FLEXAM_BENCH_ONE_DEVICE=1 FLEXAM_BENCH_BACKEND=gloo python bench.py --gpus 8 --layers 4 --steps 2 --warmup 1 --no-vae --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$1 cfg_parallel=$2 run $i:', d['check']['ok'], repr(d['check']['rel_rms_vs_single_gpu']), repr(d['check']['worst_rank_rel_rms']), d['launch']['attempt'])"; done; done

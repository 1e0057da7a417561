# r6z: the emulated rank of 2 and of 4 GPUs at HEAD (same line as the default's rank of 8, link model included)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6z
mkdir -p $O
cd $R
for n in 2 4; do
  python bench.py --steps 12 --warmup 4 --no-vae --no-cpu-baseline --emulate-rank $n > $O/bench_emulate_rank_$n.json 2> $O/bench_emulate_rank_$n.err; echo "rank-of-$n rc $?" >> $O/rc.txt
done
cat $O/rc.txt
python - <<'PY'
import json
for n in (2, 4):
    d = json.loads(open(f"gpurun_out/r6z/bench_emulate_rank_{n}.json").read().strip().splitlines()[-1])
    e = d["emulated_ranks"]
    print(n, d["ms_per_step"], e["predicted_scaling_no_comm"], e.get("predicted_scaling_at_link_GBps"))
    for r in e["layouts"]:
        print("  ", r.get("layout"), r.get("ms_per_step"), r.get("host_share_of_step"), r.get("ms_per_step_at_link_GBps"))
PY

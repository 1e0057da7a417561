# r6zl: inter-kernel idle time of one emulated rank of eight (default layout, collectives that move nothing) and of the single-GPU step:
# kernel trace with timestamps, gaps between consecutive kernels of the compute stream
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6zl
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/rank8 -- python3 $R/tools/emulate_rank.py 8 0 6 2 > $O/rank8.txt 2> $O/rank8.log
rocprofv3 --kernel-trace --output-format csv -d $O/single -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vae --no-clip --no-kernel-timing --emulate-rank 0 > $O/single.json 2> $O/single.log
cd $R
python3 - <<'PY'
import csv, glob, collections
for name in ("rank8", "single"):
    f = glob.glob(f"gpurun_out/r6zl/{name}/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    by_q = collections.defaultdict(list)
    for r in rows:
        by_q[(r["Queue_Id"], r.get("Stream_Id"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    q, ks = max(by_q.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))
    ks.sort()
    # the steady part: the last 60 % of the launches of the busiest queue
    ks = ks[int(len(ks) * 0.4):]
    busy = sum(e - s for s, e, _ in ks)
    gaps = [max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1)]
    span = ks[-1][1] - ks[0][0]
    small = [g for g in gaps if g < 50000]
    print(f"{name}: queue {q}, {len(ks)} launches, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({busy/span:.3f}), gaps total {sum(gaps)/1e6:.2f} ms; "
          f"gaps < 50 us: {len(small)} with median {sorted(small)[len(small)//2]/1e3:.2f} us mean {sum(small)/len(small)/1e3:.2f} us, sum {sum(small)/1e6:.2f} ms; larger gaps: {len(gaps)-len(small)} summing {sum(g for g in gaps if g >= 50000)/1e6:.2f} ms")
PY
cat $O/rank8.txt | tail -1

cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/kt_c5; rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python3 bench.py --workload c5 --steps 200 --warmup 20 --legs none --cpu-edges 0 --no-score --no-profile > $O/bench.json 2> $O/kt.err
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/kt_c5/kt/kt_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40% of the trace = timed region
n=len(rows); tail=rows[int(n*0.75):]
agg=collections.defaultdict(list)
for r in tail: agg[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
span=(int(tail[-1]['End_Timestamp'])-int(tail[0]['Start_Timestamp']))/1000
print("span us",span)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])): print("%-62s n %5d avg %8.1f sum %9.0f  %.1f %% of span"%(k,len(v),sum(v)/len(v),sum(v),100*sum(v)/span))
PY
rm -rf $O/kt

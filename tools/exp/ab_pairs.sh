#!/bin/bash
# paired hub hops against single ones, alternately on ONE box: C5 over 200 steps and the driver's 20, C3 over 100
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
for r in 1 2; do
  for v in pair single; do
    F=""; [ $v = pair ] && F="--chain-pairs"
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 --legs c3 --cpu-edges 0 --no-score $F > gpurun_out/r5/ab_pairs_${v}_200_$r.json 2> gpurun_out/r5/ab_pairs_${v}_200_$r.err
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --legs none --cpu-edges 0 --no-score $F > gpurun_out/r5/ab_pairs_${v}_20_$r.json 2> gpurun_out/r5/ab_pairs_${v}_20_$r.err
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/ab_pairs_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
    except Exception as e:
        print(f, 'FAILED', e); continue
    ks=d['kernels'].get('tppr_stream',{})
    print(f.split('/')[-1], 'c5 %.4f ms/step' % d['ms_per_step'], 'k_stream %.0f us x %d' % (ks.get('avg_us',0), ks.get('launches',0)), d.get('chain_hops'))
    for n,w in d.get('workloads',{}).items():
        ks=w['kernels'].get('tppr_stream',{})
        print('    ', n, '%.4f ms/step' % w['ms_per_step'], 'k_stream %.0f us' % ks.get('avg_us',0), w.get('chain_hops'))
PY

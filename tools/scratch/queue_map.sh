export TMPDIR=/tmp
for o in A; do
  tag=$(echo $o | tr -d ,)
  rocprofv3 --kernel-trace -d $PWD/gpurun_out/qmap_$tag -o t --output-format csv -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --repeats 0 --no-verify --stream-order $o > gpurun_out/qmap_$tag.json 2> gpurun_out/qmap_$tag.err || { tail -5 gpurun_out/qmap_$tag.err; exit 1; }
  python3 - $PWD/gpurun_out/qmap_$tag <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(f, len(rows), list(rows[0].keys()))
cols = [c for c in rows[0] if 'ueue' in c or 'tream' in c]
m = collections.Counter()
for r in rows:
    n = r['Kernel_Name']
    role = 'main' if 'k_trace_primary' in n else ('side' if 'k_trace_dda_walk' in n else None)
    if role:
        m[(role,) + tuple(r[c] for c in cols)] += 1
print(cols)
for k, v in sorted(m.items()):
    print(k, v)
PY
  rm -rf gpurun_out/qmap_$tag
done

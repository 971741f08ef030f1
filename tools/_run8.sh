cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/pmc_sq.sh walk32 0 32 > gpurun_out/pmc_walk32.log 2>&1
bash tools/pmc_sq.sh beam64 2 64 > gpurun_out/pmc_beam64.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for tag in ('walk32','beam64'):
    acc=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv'%tag)+glob.glob('gpurun_out/pmc_%s/*/*/*counter_collection.csv'%tag):
        for r in csv.DictReader(open(f)):
            if 'k_trace_dda' in r['Kernel_Name'] and 'ILb1' not in r['Kernel_Name'] and '<true' not in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    a={k:sum(v)/len(v) for k,v in acc.items()}
    print(tag, {k:round(v) for k,v in a.items()})
    wc=a.get('SQ_WAVE_CYCLES',1)
    print('  issuing %.2f waiting %.2f stalled %.2f ; VALU %.1fM SALU %.1fM LDS %.1fM VMEM %.1fM SMEM %.1fM' % (a.get('SQ_ACTIVE_INST_ANY',0)/wc, a.get('SQ_WAIT_ANY',0)/wc, a.get('SQ_WAIT_INST_ANY',0)/wc, a.get('SQ_INSTS_VALU',0)/1e6, a.get('SQ_INSTS_SALU',0)/1e6, a.get('SQ_INSTS_LDS',0)/1e6, a.get('SQ_INSTS_VMEM_RD',0)/1e6, a.get('SQ_INSTS_SMEM',0)/1e6))
PY

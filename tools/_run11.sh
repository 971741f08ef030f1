cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "0 64" "1 32"; do
tag=$(echo $cfg | tr ' ' '_')
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_prim_$tag/p1 -o p1 --output-format csv -- python3 tools/primary_only.py $cfg > gpurun_out/pmc_prim_$tag.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d gpurun_out/pmc_prim_$tag/p2 -o p2 --output-format csv -- python3 tools/primary_only.py $cfg >> gpurun_out/pmc_prim_$tag.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for tag in ('0_64','1_32'):
    acc=collections.defaultdict(list); dur=[]
    for f in glob.glob('gpurun_out/pmc_prim_%s/*/*counter_collection.csv'%tag):
        for r in csv.DictReader(open(f)):
            if 'k_trace_primary' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob('gpurun_out/pmc_prim_%s/p1/*kernel_trace.csv'%tag):
        for r in csv.DictReader(open(f)):
            if 'k_trace_primary' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    a={k:sum(v)/len(v) for k,v in acc.items()}
    wc=a.get('SQ_WAVE_CYCLES',1)
    print(tag, 'us', [round(x) for x in dur], 'waves', a.get('SQ_WAVES'), 'wavecycles(quad) %.1fM busy %.1fM' % (wc/1e6, a.get('SQ_BUSY_CYCLES',0)/1e6))
    print('  issuing %.2f waiting %.2f stalled %.2f ldsstall %.3f; VALU %.1fM SALU %.1fM LDS %.1fM VMEM %.1fM SMEM %.1fM bankconf %.1fM' % (a.get('SQ_ACTIVE_INST_ANY',0)/wc, a.get('SQ_WAIT_ANY',0)/wc, a.get('SQ_WAIT_INST_ANY',0)/wc, a.get('SQ_WAIT_INST_LDS',0)/wc, a.get('SQ_INSTS_VALU',0)/1e6, a.get('SQ_INSTS_SALU',0)/1e6, a.get('SQ_INSTS_LDS',0)/1e6, a.get('SQ_INSTS_VMEM_RD',0)/1e6, a.get('SQ_INSTS_SMEM',0)/1e6, a.get('SQ_LDS_BANK_CONFLICT',0)/1e6))
PY

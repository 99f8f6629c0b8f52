cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02s
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES -d gpurun_out/r02s/pmc1 --output-format csv -- python3 bench_configs.py q1_packed > gpurun_out/r02s/q1.out 2> gpurun_out/r02s/q1.err; echo rc=$?
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES -d gpurun_out/r02s/pmc2 --output-format csv -- python3 bench_configs.py q1_packed > gpurun_out/r02s/q1b.out 2> gpurun_out/r02s/q1b.err; echo rc=$?
python3 - <<'PY'
import csv,glob,collections
for d in ('pmc1','pmc2'):
    f=glob.glob('gpurun_out/r02s/%s/*/*counter_collection.csv'%d)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'k_group_sum' in r['Kernel_Name']:
            acc['g'][r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    print(d, {k:v/n[k] for k,v in acc['g'].items()}, dict(n))
PY

# FETCH_SIZE (x2 per the calibration in profiles/r02_traffic.json) of the gather and tile edge kernels on the n320 processor graph
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_edge -o fetch --output-format csv -- python3 tools/lab_edge_mfma/lab.py n320_ico6 > gpurun_out/pmc_edge_run.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_edge/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if r['Counter_Name']=='FETCH_SIZE' and 'edge' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, len(v), 'mean FETCH_SIZE kb', sum(v)/len(v), '-> MB read', 2*sum(v)/len(v)/1024)
PY

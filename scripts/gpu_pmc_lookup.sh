#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in new:blackstar_amd/libblackstar_gpu.so:synthetic prev:variants_prev.so:synthetic nostars:blackstar_amd/libblackstar_gpu.so:none; do
  IFS=: read name lib stars <<< "$v"
  BLACKSTAR_LIB=$PWD/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmcl_$name -o sq -- python scripts/prof_frame.py --mode fast --frames 3 --stars $stars > gpurun_out/pmcl_$name.log 2>&1
  BLACKSTAR_LIB=$PWD/$lib rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcg_$name -o g -- python scripts/prof_frame.py --mode fast --frames 3 --stars $stars > gpurun_out/pmcg_$name.log 2>&1
done
python - <<'PY'
import csv, collections
for name in ('new','prev','nostars'):
    out={}
    for d,f in ((f'gpurun_out/pmcl_{name}','sq'),(f'gpurun_out/pmcg_{name}','g')):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f'{d}/{f}_counter_collection.csv')):
            if 'trace_frame' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
        out.update({k:sum(v)/len(v) for k,v in agg.items()})
        ds=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in csv.DictReader(open(f'{d}/{f}_kernel_trace.csv')) if 'trace_frame' in r['Kernel_Name']]
        out['ns_'+f]=sum(ds)/len(ds)
    print(name, {k:round(v/1e6,2) for k,v in out.items()})
PY

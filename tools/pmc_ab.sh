#!/bin/bash
# tools/pmc_ab.sh <variant|main> <map> -- run on the GPU box (gpurun): SQ counters (two passes) and kernel stats of the
# first sweep pass for the library in the tree (main) or a variant build variants/<variant>.so (MCL_LIB), printed as
# one line each: the A/B tool behind the round-3 kernel work (VALU wave-instructions, lane utilisation, waits, LDS).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=$1; M=$2
if [ $V != main ]; then export MCL_LIB=$R/variants/$V.so; else unset MCL_LIB; fi
O=/tmp/pmc_${V}_${M}; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $O/a -- python3 $R/bench.py --map $M --steps 4 --warmup 1 --only-main > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/b -- python3 $R/bench.py --map $M --steps 4 --warmup 1 --only-main > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 $R/bench.py --map $M --steps 20 --warmup 3 --only-main > /dev/null 2>&1
python3 - $O $V $M <<'PY'
import csv,glob,sys,os
O,V,M=sys.argv[1:4]
acc={}
for p in glob.glob(O+'/[ab]/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if r['Kernel_Name'].startswith('void k_mbes_sweep') and ', false, false>' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'],{}).setdefault(r['Dispatch_Id'],0.0)
            acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
m={k:sum(v.values())/len(v) for k,v in acc.items()}
print(V,M,{k:'%.4g'%v for k,v in sorted(m.items())})
if 'SQ_THREAD_CYCLES_VALU' in m: print(' lane util %.3f  wait_any/wave_cycles %.3f  valu/lds insts %.1f' % (m['SQ_THREAD_CYCLES_VALU']/(64*m['SQ_ACTIVE_INST_VALU']), m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'], m['SQ_INSTS_VALU']/max(m['SQ_INSTS_LDS'],1)))
for p in glob.glob(O+'/s/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if 'sweep' in r['Name'] or 'predict' in r['Name'] or 'gather' in r['Name'] or 'expand' in r['Name'] or 'quantise' in r['Name']:
            print('  ',r['Name'][:60],r['Calls'],r['AverageNs'])
PY

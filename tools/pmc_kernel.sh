#!/bin/bash
# tools/pmc_kernel.sh <kernel name prefix> <bench.py args ...> -- run on the GPU box (gpurun): SQ counters (two passes)
# of one kernel of a bench run, printed as one line: VALU wave-instructions, lane utilisation, waits, LDS.
# e.g.  tools/pmc_kernel.sh 'void k_mbes_slice<false>' --map mesh-soup
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
O=/tmp/pmc_k; rm -rf $O; mkdir -p $O
# PMC_PROG="tools/experiments/leg.py config5 6": another program of the repo instead of bench.py's main leg
if [ -n "$PMC_PROG" ]; then set -- $R/$PMC_PROG; else set -- $R/bench.py "$@" --steps 4 --warmup 1 --only-main; fi
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $O/a -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/b -- python3 "$@" > /dev/null 2>&1
python3 - $O "$K" <<'PY'
import csv,glob,sys
O,K=sys.argv[1:3]
acc={}
for p in glob.glob(O+'/[ab]/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if r['Kernel_Name'].startswith(K):
            acc.setdefault(r['Counter_Name'],{}).setdefault(r['Dispatch_Id'],0.0)
            acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
m={k:sum(v.values())/len(v) for k,v in acc.items()}
print(K,{k:'%.4g'%v for k,v in sorted(m.items())})
if 'SQ_THREAD_CYCLES_VALU' in m: print(' lane util %.3f  wait_any/wave_cycles %.3f  valu/lds insts %.1f  valu per wave %.0f  lds per wave %.0f' % (m['SQ_THREAD_CYCLES_VALU']/(64*m['SQ_ACTIVE_INST_VALU']), m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'], m['SQ_INSTS_VALU']/max(m['SQ_INSTS_LDS'],1), m['SQ_INSTS_VALU']/m['SQ_WAVES'], m['SQ_INSTS_LDS']/m['SQ_WAVES']))
PY

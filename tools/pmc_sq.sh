#!/bin/bash
# SQ counters for the dominant kernels (own pass, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-extras > $OUT/a.log 2>&1
echo "rc=$?" >> $OUT/a.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-extras > $OUT/b.log 2>&1
echo "rc=$?" >> $OUT/b.log
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/c -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-profile --no-extras > $OUT/c.log 2>&1
echo "rc=$?" >> $OUT/c.log
tail -3 $OUT/c.log
rocprofv3 -L 2>/dev/null | grep -o "\(TA_\|TCP_\|SQ_INST_CYCLES\|SQ_VALU\)[A-Z0-9_a-z]*" | sort -u | tr '\n' ' ' | cut -c1-3000
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
for tag in ('a','b','c'):
    fs=glob.glob(f'{R}/gpurun_out/pmc_sq/{tag}/*/*_counter_collection.csv')
    if not fs: print(tag,'no output'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
    for r in csv.DictReader(open(fs[0])):
        k=(r['Kernel_Name'][:60], r['Grid_Size'])
        a=acc[k][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    rows=sorted(acc.items(), key=lambda kv:-max(v[0] for v in kv[1].values()))[:8]
    for k,d in rows:
        print(k, {c: round(v[0]/v[1]) for c,v in d.items()})
PY

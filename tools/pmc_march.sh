#!/bin/bash
# per-dispatch SQ counters of the marching kernel over single frames (one frame in flight)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm
mkdir -p $OUT
KERNEL=${KERNEL:-march_frame}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py ${SCENE:-dnerf} > $OUT.p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(dict)
for p in (1,2):
    f=glob.glob("$OUT/p%d/*counter_collection.csv"%p)[0]
    for r in csv.DictReader(open(f)):
        if "$KERNEL" not in r["Kernel_Name"]: continue
        k=int(r["Dispatch_Id"])
        d[(p,k)][r["Counter_Name"]]=d[(p,k)].get(r["Counter_Name"],0)+float(r["Counter_Value"])
for p in (1,2):
    ks=sorted(k for (pp,k) in d if pp==p)[:13]
    print("pass",p)
    for k in ks:
        c=d[(p,k)]
        print("  ",k," ".join("%s=%.3g"%(n.replace("SQ_",""),v) for n,v in sorted(c.items())))
PY

#!/bin/bash
# microbench timing + per-launch instruction counts (GPU box)
cd "$(dirname "$0")/.."
python tools/microbench_fused.py "$@" 2>&1 | grep -v amdgpu.ids > /tmp/mb_time.txt
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/mb && rocprofv3 --kernel-trace --output-format csv --kernel-include-regex fused_pass --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/mb -o mb -- python $OLDPWD/tools/microbench_fused.py --reps 1 "$@" > /dev/null 2>&1)
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/mb/**/*counter_collection.csv",recursive=True)[0]
d=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
rows=[v for k,v in d.items()][1::2]
lines=[l.rstrip() for l in open("/tmp/mb_time.txt") if l.strip()]
print(lines[0])
for l,v in zip(lines[1:],rows):
    w=v["SQ_WAVES"]
    print("%s | VALU %5.0f SALU %5.0f LDS %4.0f conf %.2f"%(l, v["SQ_INSTS_VALU"]/w, v["SQ_INSTS_SALU"]/w, v["SQ_INSTS_LDS"]/w, v["SQ_LDS_BANK_CONFLICT"]/max(v["SQ_LDS_IDX_ACTIVE"],1)))
PY

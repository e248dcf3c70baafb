# per-kernel durations of tools/experiments/prof_reductions.py (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_red -o red -- python $R/tools/experiments/prof_reductions.py > /tmp/prof_red.log 2>&1 || tail -20 /tmp/prof_red.log
python - <<PY | tee $R/gpurun_out/prof_reductions.txt
import csv, glob, collections
fs = glob.glob("/tmp/prof_red/**/*kernel_trace.csv", recursive=True)
print(fs)
d = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    d[(r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in d.items():
    print(f"{k[0]:62s} grid {k[1]:>9s} n={len(v):3d} min {min(v):7.3f} med {sorted(v)[len(v)//2]:7.3f} ms")
PY

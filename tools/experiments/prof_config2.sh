# kernel durations of BASELINE config 2 (n = 24, depth 20, complex128): is the step host- or GPU-bound?
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_config2.py 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o c2 -- python $R/tools/bench_config2.py --reps 20 > /tmp/prof_c2.log 2>&1 || tail -5 /tmp/prof_c2.log
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/prof_c2/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:72s} n={len(v):4d} avg {sum(v)/len(v):8.1f} us  total {sum(v)/23/1e3:7.3f} ms/step")
    tot += sum(v)
print("GPU busy per step (23 steps):", tot / 23 / 1e3, "ms")
PY

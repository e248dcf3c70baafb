#!/bin/bash
# Everything profiles/<tag>/ holds, in one go on the GPU box:  gpurun -- 'bash tools/refresh_profiles.sh r02'
# then copy gpurun_out/prof/<tag>/{summary.txt,kernel_stats.csv,traffic_*.json,bench_*.json,*.txt} to profiles/<tag>/.
cd "$(dirname "$0")/.."
tag=${1:-r06}
root=$PWD/gpurun_out/prof/$tag
bash tools/profile.sh "$tag" > /dev/null 2>&1
cp "$root/traffic.json" "$root/traffic_n28_b16_c64.json" 2>/dev/null
cp "$root"/trace/*kernel_stats.csv "$root/kernel_stats.csv" 2>/dev/null || find "$root/trace" -name '*kernel_stats.csv' -exec cp {} "$root/kernel_stats.csv" \;
cp "$root/bench_trace.json" "$root/bench_under_rocprof.json"
python bench.py --traffic-json "$root/traffic_n28_b16_c64.json" > "$root/bench_default.json" 2> "$root/bench_default.err"
bash tools/ablation_table.sh > "$root/ablation.txt" 2>&1
bash tools/mb_counters.sh > "$root/microbench.txt" 2>&1
{
  python tools/bench_train.py --n 20 --depth 20 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 20 --depth 20 --modes adjoint --no-fused-sweep 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 24 --depth 20 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 24 --depth 20 --modes adjoint --no-fused-sweep 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids
  echo '# A/B: every reduction record forms all of G (DQ_REDUCED_GRAD=0)'
  DQ_REDUCED_GRAD=0 python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids
  DQ_REDUCED_GRAD=0 python tools/bench_train.py --n 27 --depth 40 --modes adjoint --dtype c128 2>&1 | grep -v amdgpu.ids
  echo '# A/B: the sweep of rounds 3-5a -- at most 72 gates + reductions per pass (DQ_MAX_GATES=72), two sums per X rotation (DQ_TERMINAL_GRAD=0)'
  DQ_MAX_GATES=72 DQ_TERMINAL_GRAD=0 python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 28 --depth 40 --modes adjoint --no-fused-sweep 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 24 --depth 20 --modes adjoint --dtype c128 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 24 --depth 20 --modes adjoint --dtype c128 --no-fused-sweep 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 27 --depth 40 --modes adjoint --dtype c128 2>&1 | grep -v amdgpu.ids
  python tools/bench_train.py --n 27 --depth 40 --modes adjoint --dtype c128 --no-fused-sweep 2>&1 | grep -v amdgpu.ids
  python tools/dump_sweep_passes.py 2>&1 | grep -v amdgpu.ids
  python tools/bench_small.py 2>&1 | grep -v amdgpu.ids
  python tools/bench_density.py 2>&1 | grep -v amdgpu.ids
  echo '# ... from a state that is not |0><0| (every pass moves the whole matrix); then with the real 4x4 bodies of round 4 in place of the X-shaped ones'
  python tools/bench_density.py --no-zero-state 2>&1 | grep -v amdgpu.ids
  python tools/bench_density.py --no-zero-state --real-bodies 2>&1 | grep -v amdgpu.ids
  python tools/bench_expect.py 2>&1 | grep -v amdgpu.ids
  python tools/bench_config2.py --cpu 2>&1 | grep -v amdgpu.ids
  python tools/bench_single_gate_kernel.py 2>&1 | grep -v amdgpu.ids
  python bench.py --dtype c128 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null
  echo "# the one-GPU anchor of the strong-scaling pair (SURVEY 8d): n = 31, batch 1"
  python bench.py --strong --steps 3 --warmup 1 --no-cpu-baseline --no-sweep 2>/dev/null
} > "$root/secondary_benchmarks.txt" 2>&1
{
  for a in "--batch 4" "--strong" "--config 4" "--config 5" "--batch 4 --batch-shard"; do
    echo "# python bench.py --gpus 2 --backend gloo --nqubit 24 --depth 10 $a (self-launched; two gloo ranks sharing one GPU: functional check, not a performance number)"
    python bench.py --gpus 2 --backend gloo --steps 1 --warmup 1 --nqubit 24 --depth 10 --no-cpu-baseline --no-sweep $a 2>&1 | grep '^{' | tail -1
  done
} > "$root/two_ranks_one_gpu_functional.txt" 2>&1
{
  echo "# bench.py --rehearse-rank R: the COMPUTE half of rank R's step of the strong-scaling job (n = 34 on 8 ranks), measured on this one GPU"
  for r in 0 1; do for v in 0 2; do
    python bench.py --gpus 8 --strong --rehearse-rank $r --virtual-bits $v --steps 5 --warmup 1 2>/dev/null | grep '^{'
  done; done
  echo "# ... and of config 4 (n = 32 on 4 ranks), rank 0 and rank 1"
  for r in 0 1; do python bench.py --gpus 4 --config 4 --rehearse-rank $r --virtual-bits 0 --steps 5 --warmup 1 2>/dev/null | grep '^{'; done
} > "$root/strong_rehearsal.txt" 2>&1
python tools/bench_vmap.py 20 16 2>&1 | grep -v amdgpu.ids > "$root/bench_vmap.txt"
python tools/bench_vmap.py 16 8 hessian 2>&1 | grep -v amdgpu.ids > "$root/bench_vmap_hessian_n16.txt"
python tools/bench_dense.py 2>&1 | grep -v amdgpu.ids > "$root/bench_dense.txt"
python tools/dump_passes.py 2>&1 | grep -v amdgpu.ids > "$root/passes_headline.txt"
python tools/bench_gradient_reference.py --trials 3 2>&1 | grep -v "amdgpu.ids\|UserWarning\|run_backward" > "$root/bench_gradient_reference.txt"
python tools/crosscheck_large.py 2>&1 | grep -v amdgpu.ids > "$root/crosscheck_large.txt"
python tools/experiments/bench_shard_helpers.py 2>&1 | grep -v amdgpu.ids > "$root/shard_helpers_and_reductions.txt"
ls -la "$root"

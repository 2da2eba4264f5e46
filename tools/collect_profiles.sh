#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace summaries and FETCH_SIZE / WRITE_SIZE passes (separate
# rocprofv3 invocations, the program directly after `--`).  Output under gpurun_out/$1/.
set -o pipefail
OUT=gpurun_out/${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
KERN="rollout rollout64k rollout512k synth step record return episode reward assemble nan reset"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace_bench.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_headline" -- python3 bench.py --no-extras --no-cpu-baseline > "$OUT/bench_headline_under_rocprof.json" 2> "$OUT/trace_headline.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_kernels" -- python3 tools/prof_kernels.py $KERN > "$OUT/trace_kernels.log" 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  s=$(echo $c | cut -c1 | tr 'FW' 'fw'); [ $c = FETCH_SIZE ] && s=fetch || s=write
  rocprofv3 --pmc $c --output-format csv -d "$OUT/cal_$s" -- ./tools/pmc_calibrate > "$OUT/cal_$s.log" 2>&1 || exit 1
  rocprofv3 --pmc $c --output-format csv -d "$OUT/k_$s" -- python3 tools/prof_kernels.py $KERN > "$OUT/k_$s.log" 2>&1 || exit 1
  rocprofv3 --pmc $c --output-format csv -d "$OUT/b_$s" -- python3 bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > "$OUT/b_$s.json" 2> "$OUT/b_$s.err" || exit 1
done
echo collected

#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace summaries and FETCH_SIZE / WRITE_SIZE passes (separate
# rocprofv3 invocations, the program directly after `--`).  Output under gpurun_out/$1/.
# The kernels run in two groups so that every launch shape is told apart by (kernel name, grid size): the
# 524 288-env and the 1 M-env rollouts are the same kernel on the same grid.
set -o pipefail
OUT=gpurun_out/${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
KERN="rollout rollout64k rollout128k rollout256k synth pg step record return scan episode reward assemble nan reset trim multi64k rolloutret contacts"
KERN2="rollout512k rollout4m multi1m rolloutret512k"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench" -- python3 bench.py --no-cpu-baseline --no-box-state > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace_bench.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_headline" -- python3 bench.py --no-extras --no-cpu-baseline --no-stride-probe --no-box-state > "$OUT/bench_headline_under_rocprof.json" 2> "$OUT/trace_headline.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_kernels" -- python3 tools/prof_kernels.py $KERN > "$OUT/trace_kernels.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_kernels2" -- python3 tools/prof_kernels.py $KERN2 > "$OUT/trace_kernels2.log" 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  [ $c = FETCH_SIZE ] && s=fetch || s=write
  rocprofv3 --pmc $c --output-format csv -d "$OUT/cal_$s" -- ./tools/pmc_calibrate > "$OUT/cal_$s.log" 2>&1 || exit 1
  rocprofv3 --pmc $c --output-format csv -d "$OUT/k_$s" -- python3 tools/prof_kernels.py $KERN > "$OUT/k_$s.log" 2>&1 || exit 1
  rocprofv3 --pmc $c --output-format csv -d "$OUT/k2_$s" -- python3 tools/prof_kernels.py $KERN2 > "$OUT/k2_$s.log" 2>&1 || exit 1
  rocprofv3 --pmc $c --output-format csv -d "$OUT/b_$s" -- python3 bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-stride-probe --no-box-state > "$OUT/b_$s.json" 2> "$OUT/b_$s.err" || exit 1
done
echo collected

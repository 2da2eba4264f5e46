#!/bin/bash
# Copies the summaries of a tools/collect_profiles.sh run (gpurun_out/$1) into profiles/ under the
# round prefix $2 (default r02): kernel-stat CSVs, the bench lines printed under the tracer, raw PMC
# counter CSVs and the per-kernel traffic / mean-duration JSONs (tools/summarize_pmc.py).
set -e
SRC=gpurun_out/${1:?run directory under gpurun_out/}
R=${2:-r05}
cd "$(dirname "$0")/.."
one() {      # exactly one result file per sub-run (gpurun MERGES into an existing local directory: use a fresh name per run)
  n=$(find "$SRC/$1" -name "*$2" | wc -l)
  [ "$n" = 1 ] || { echo "$SRC/$1: $n files match *$2 (stale files from an earlier run of the same name?)" >&2; exit 1; }
  find "$SRC/$1" -name "*$2"
}
cp "$(one trace_headline kernel_stats.csv)" profiles/${R}_kernel_stats_headline.csv
cp "$(one trace_bench kernel_stats.csv)" profiles/${R}_kernel_stats_bench_all.csv
cp "$(one trace_kernels kernel_stats.csv)" profiles/${R}_kernel_stats_kernels.csv
cp "$(one trace_kernels2 kernel_stats.csv)" profiles/${R}_kernel_stats_kernels_512k_4m.csv
tail -n 1 "$SRC/bench_headline_under_rocprof.json" > profiles/${R}_bench_headline_under_rocprof.json
tail -n 1 "$SRC/bench_under_rocprof.json" > profiles/${R}_bench_all_under_rocprof.json
mkdir -p profiles/${R}_pmc
for s in fetch write; do
  cp "$(one cal_$s counter_collection.csv)" profiles/${R}_pmc/cal_${s}_counter_collection.csv
  cp "$(one k_$s counter_collection.csv)" profiles/${R}_pmc/kernels_${s}_counter_collection.csv
  cp "$(one k2_$s counter_collection.csv)" profiles/${R}_pmc/kernels_512k_4m_${s}_counter_collection.csv
  cp "$(one b_$s counter_collection.csv)" profiles/${R}_pmc/b_${s}_counter_collection.csv
done
python tools/summarize_pmc.py "$SRC" profiles/${R}_pmc > /dev/null
mv profiles/${R}_pmc/traffic.json profiles/traffic.json
echo published "$SRC" as profiles/${R}_*

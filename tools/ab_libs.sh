#!/bin/bash
# A/B of library variants on ONE box: the headline launch (bench.py, HIP events) with BPGYM_LIB pointing at each variant in turn, twice,
# alternating.  Variants: build.build(force=True, extra_flags=['-D...'], out='tools/libbpgym_<name>.so') in the build container.
#   gpurun -- ./tools/ab_libs.sh tools/libbpgym_stagger1.so tools/libbpgym_stagger4.so
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for lib in "" "$@"; do
    BPGYM_LIB="${lib:+$PWD/$lib}" python3 bench.py --no-extras --no-cpu-baseline --no-stride-probe --no-box-state --no-sharded --steps 100 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s kernel %.2f us  frac %.3f  spread %s' % ('${lib:-product}', d['roofline']['kernel_ms']*1e3, d['roofline']['frac'], [round(x*1e3,1) for x in d['kernel_ms_spread']['later_groups']]))"
  done
done

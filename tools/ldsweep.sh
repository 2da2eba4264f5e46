mkdir -p gpurun_out/r02
: > gpurun_out/r02/tune_ld2.log
for pad in 0 32 64 128 192 256 320 384 512 768 1024 1280 1536 2048 2304 3072 3840 4352 8448; do
  LDPAD=$pad GEO_SHORT=2 timeout -k 10 200 ./tools/tune_geometry 1048576 4194304 >> gpurun_out/r02/tune_ld2.log 2>&1 || exit 1
done

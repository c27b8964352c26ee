source scripts/dev/r5/pool_bench.sh r5n 300 20
run g4_rr_300 "--groups 4" A=1
run g8_pool_300 "--groups 8 --pool 1 --threads 1" FSIM_BENCH_TRACE=1
grep "tables for slab" gpurun_out/r5n/g8_pool_300.err | head -4
run g16_pool_300 "--groups 16 --pool 1 --threads 1" A=1

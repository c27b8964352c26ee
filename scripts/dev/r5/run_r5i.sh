source scripts/dev/r5/pool_bench.sh r5i 20 5
run g8_pool_20 "--groups 8 --pool 1 --threads 1" A=1
run g4_legacy_20 "--groups 4" A=1
run g8_legacy_thr_20 "--groups 8 --threads 1" GPU_MAX_HW_QUEUES=8
source scripts/dev/r5/pool_bench.sh r5i 60 5
run g8_pool_60_nola "--groups 8 --pool 1 --threads 1 --no-lookahead" A=1
run g8_pool_60 "--groups 8 --pool 1 --threads 1" A=1
run g8_pool_60_age1 "--groups 8 --pool 1 --threads 1" FSIM_POOL_JOB_AGE_MS=1.0

source scripts/dev/r5/pool_bench.sh r5f 60 5
run g4_pool "--groups 4 --pool 1 --threads 1" A=1
run g8_pool "--groups 8 --pool 1 --threads 1" A=1
run g16_pool "--groups 16 --pool 1 --threads 1" A=1
run g16_pool_t64 "--groups 16 --pool 1 --threads 1" FSIM_POOL_TEAMS=64
run g16_pool_t256 "--groups 16 --pool 1 --threads 1" FSIM_POOL_TEAMS=256
run g8_pool_t256 "--groups 8 --pool 1 --threads 1" FSIM_POOL_TEAMS=256
run g16_pool_t495 "--groups 16 --pool 1 --threads 1" FSIM_POOL_TEAMS=495

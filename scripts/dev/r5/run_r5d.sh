source scripts/dev/r5/pool_bench.sh r5d 60 5
run g4_legacy "--groups 4" A=1
run g4_pool "--groups 4 --pool 1 --threads 1" A=1
run g8_pool "--groups 8 --pool 1 --threads 1" A=1
run g16_pool "--groups 16 --pool 1 --threads 1" A=1
run g16_pool_nothr "--groups 16 --pool 1" A=1
run g32_pool "--groups 32 --pool 1 --threads 1" A=1

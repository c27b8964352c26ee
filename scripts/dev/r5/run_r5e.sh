source scripts/dev/r5/pool_bench.sh r5e 60 5
run g4_pool "--groups 4 --pool 1 --threads 1" A=1
run g8_pool "--groups 8 --pool 1 --threads 1" A=1
run g16_pool "--groups 16 --pool 1 --threads 1" A=1
run g4_pool_nofence "--groups 4 --pool 1 --threads 1" FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_nofence.so
run g16_pool_nofence "--groups 16 --pool 1 --threads 1" FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_nofence.so
run g16_pool_t32 "--groups 16 --pool 1 --threads 1" FSIM_POOL_TEAMS=32
run g16_pool_t96 "--groups 16 --pool 1 --threads 1" FSIM_POOL_TEAMS=96

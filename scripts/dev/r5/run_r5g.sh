source scripts/dev/r5/pool_bench.sh r5g 60 5
run g8_pool "--groups 8 --pool 1 --threads 1" A=1
run g8_pool_nofence "--groups 8 --pool 1 --threads 1" FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_nofence.so
run g8_pool_norel "--groups 8 --pool 1 --threads 1" FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_norel.so
run g16_pool_nofence "--groups 16 --pool 1 --threads 1" FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_nofence.so
run g4_pool_nofence "--groups 4 --pool 1 --threads 1" FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_nofence.so

source scripts/dev/r5/pool_bench.sh r5m 20 5
for rep in 1 2 3; do
run g4_rr_$rep "--groups 4" A=1
run g4_thr_$rep "--groups 4 --threads 1" A=1
done
source scripts/dev/r5/pool_bench.sh r5m 300 20
run g4_rr_300 "--groups 4" A=1
run g4_thr_300 "--groups 4 --threads 1" A=1
run g8_pool_300 "--groups 8 --pool 1 --threads 1" A=1
run g16_pool_300 "--groups 16 --pool 1 --threads 1" A=1

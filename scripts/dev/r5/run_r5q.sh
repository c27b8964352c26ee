source scripts/dev/r5/pool_bench.sh r5q 20 5
for rep in 1 2; do
for K in 100 125 150 200; do
run k${K}_$rep "--groups 4 --episode-window 0" FSIM_MW_K=$K
done
run k150_thr_$rep "--groups 4 --threads 1 --episode-window 0" FSIM_MW_K=150
run k125_thr_$rep "--groups 4 --threads 1 --episode-window 0" FSIM_MW_K=125
done

source scripts/dev/r5/pool_bench.sh r5o 300 20
for rep in 1 2 3; do run g16_pool_300_$rep "--groups 16 --pool 1 --threads 1" FSIM_BENCH_WATCHDOG=40; done
run g8_pool_300 "--groups 8 --pool 1 --threads 1" FSIM_BENCH_WATCHDOG=40
grep -l "Thread 0x" gpurun_out/r5o/*.err | head; for f in $(grep -l "Thread 0x" gpurun_out/r5o/*.err | head -1); do grep -A6 "Thread 0x\|Current thread" $f | grep "File" | grep -v site-packages | sort | uniq -c | sort -rn | head -12; done

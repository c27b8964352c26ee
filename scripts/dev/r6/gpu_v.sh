#!/bin/bash
# round 6: compiler-flag variants of the library (-Os, -O2 against -O3) on the driver command and the 100-step window
O=gpurun_out/r6v; mkdir -p $O
for i in 1 2; do
  for v in "" _os _o2 _unr _inl _rm; do  # (libfsim_<v>.so: built by hand with the flag under test, not kept)
    FSIM_LIB=$PWD/furniture_amd/csrc/libfsim$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver   lib$v', round(d['value']), round(d['ms_per_step'],3))" >> $O/ab.txt
    FSIM_LIB=$PWD/furniture_amd/csrc/libfsim$v.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps100 lib$v', round(d['value']), round(d['ms_per_step'],3))" >> $O/ab.txt
  done
done
cat $O/ab.txt

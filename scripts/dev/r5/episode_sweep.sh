#!/bin/bash
# development: the whole-episode agreement run (episode_agreement.py) across kernel paths and reset cadences -- a bug hunt, not a benchmark
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/sweep; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 200 python $R/scripts/dev/r5/episode_agreement.py $N $T $S > $O/$tag.txt 2>&1
  echo "$tag: steps $(grep -c '^t ' $O/$tag.txt)  done-eq-all $(grep -c 'done eq True' $O/$tag.txt)  info-differ $(grep -c differ $O/$tag.txt)  worst after reset $(grep -o 'after reset: max [0-9.e+-]*' $O/$tag.txt | awk '{print $5}' | sort -g | tail -1)  errors $(grep -ci 'error\|Traceback' $O/$tag.txt)"; }
N=256 T=30 S=62; run mw_off FSIM_MW=0; run mw_all FSIM_MW=all; run mw_rule FSIM_MW=1; run mw_rule_k0 FSIM_MW=1 FSIM_MW_K=0; run no_lookahead FSIM_NO_LOOKAHEAD=1; run generic FSIM_GENERIC=1
N=1024 T=6 S=40; run short_episodes_1024 A=1; run short_episodes_1024_la_chunk FSIM_LA_CHUNK=50 FSIM_LA_DEFER=0
N=2048 T=30 S=32; run batch_2048 A=1

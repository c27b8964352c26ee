#!/bin/bash
# round 6: the profile set of the committed library (rocprofv3 kernel trace + PMC passes) and the round's bench lines
bash scripts/profile_round.sh r6e > gpurun_out/r6e_profile.log 2>&1
tail -12 gpurun_out/r6e_profile.log
bash scripts/dev/r6/bench_lines.sh r6e_lines

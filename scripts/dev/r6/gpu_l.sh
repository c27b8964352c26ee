#!/bin/bash
# round 6: the divergence control on the other BASELINE configs' models
O=gpurun_out/r6l; mkdir -p $O
python scripts/divergence_control.py 256 30 62 Sawyer swivel_chair_0700 > $O/divergence_control_sawyer_swivel_chair_0700.txt 2> $O/a.err; tail -3 $O/divergence_control_sawyer_swivel_chair_0700.txt | cut -c1-250
python scripts/divergence_control.py 256 30 62 Baxter desk_mikael_1064 > $O/divergence_control_baxter_desk_mikael_1064.txt 2> $O/b.err; tail -3 $O/divergence_control_baxter_desk_mikael_1064.txt | cut -c1-250
python scripts/divergence_control.py 256 30 62 Sawyer toy_table > $O/divergence_control_sawyer_toy_table.txt 2> $O/c.err; tail -3 $O/divergence_control_sawyer_toy_table.txt | cut -c1-250

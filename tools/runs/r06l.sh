#!/bin/bash
# lease r06l: the FINAL tree -- whole GPU suite with its parity margins, smoke(), the driver's bench command, then the profile set
cd $GRAFT_REPO_ROOT
bash tools/runs/r06f.sh > gpurun_out/r06f_final.log 2>&1
tail -12 gpurun_out/r06f_final.log | cut -c1-300
bash tools/profile_r06.sh > gpurun_out/prof_r06.log 2>&1
tail -5 gpurun_out/prof_r06.log | cut -c1-200
cat gpurun_out/prof_r06/ab_1080p.txt

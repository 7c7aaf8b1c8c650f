#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r03ak; mkdir -p $O
bash tools/profile_r03.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
bash tools/runs/r03ac.sh

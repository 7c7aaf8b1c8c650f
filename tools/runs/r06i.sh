#!/bin/bash
# lease r06i: the round-6 profile set on the committed tree (tools/profile_r06.sh -> gpurun_out/prof_r06 -> tools/publish_r06.sh)
cd $GRAFT_REPO_ROOT
bash tools/profile_r06.sh > gpurun_out/prof_r06.log 2>&1
tail -60 gpurun_out/prof_r06.log | cut -c1-220

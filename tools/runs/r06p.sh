#!/bin/bash
# lease r06p: last check of the committed tree -- smoke(), the frame tests, the driver's bench command
cd $GRAFT_REPO_ROOT
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_frame.py -q -x -m gpu 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | cut -c1-330

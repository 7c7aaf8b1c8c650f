#!/bin/bash
# round 3, first GPU pass: full -m gpu suite, headline benches, configs[4] stress bank
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
timeout 600 python bench.py > $O/bench_1080p.json 2> $O/bench_1080p.err; tail -c 600 $O/bench_1080p.json
timeout 600 python bench.py --height 480 --width 832 --steps 47 --warmup 3 > $O/bench_480p.json 2> $O/bench_480p.err; tail -c 300 $O/bench_480p.json
timeout 900 python bench.py --height 2160 --width 3840 --steps 197 --warmup 3 --stress-bank --no-cpu-baseline --per-frame-report $O/4k_T200_perframe.json > $O/bench_4k_T200_growing.json 2> $O/bench_4k.err; tail -c 600 $O/bench_4k_T200_growing.json; tail -5 $O/bench_4k.err

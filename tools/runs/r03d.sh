#!/bin/bash
# round 3: memory-read PMC on the growing bank, kernel stats of the batched 480p step, check of the downsample-GN fold
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sequence_vs_oracle or two_frames or batched" > $O/pytest_fold.log 2>&1; echo "rc $?" >> $O/pytest_fold.log; tail -4 $O/pytest_fold.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_1080p.json 2> $O/bench_1080p.err
python -c "import json;a=json.load(open('$O/bench_1080p.json'));print('1080p: %.2f fps, %.2f ms, conv frac %.3f, conv ms %.2f' % (a['value'],a['ms_per_step'],a['roofline']['frac'],a['roofline']['conv_ms_per_frame']))"
timeout 1500 bash tools/memread_pmc.sh > $O/memread_growing_bank_pmc.md 2> $O/memread_pmc.err; cat $O/memread_growing_bank_pmc.md; tail -3 $O/memread_pmc.err
cd /tmp
export OTVM_TUNE_FILE=$O/tune.json
python $R/bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline --batch 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks480b4 -o ks -- python $R/bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline --batch 4 > $O/ks480b4.log 2>&1
cd $R
python tools/kernel_stats_md.py $O/ks480b4/*/ks_kernel_stats.csv 50 "python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline --batch 4" > $O/kernel_stats_480p_b4.md 2>/dev/null || python tools/kernel_stats_md.py $O/ks480b4/ks_kernel_stats.csv 50 "python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline --batch 4" > $O/kernel_stats_480p_b4.md
rm -rf $O/ks480b4/*kernel_trace.csv $O/ks480b4/*/*kernel_trace.csv
head -40 $O/kernel_stats_480p_b4.md

#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py -m gpu -x -q -k "ppm or sequence_vs_oracle or batched_sequences or fuzz_short" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
export OTVM_TUNE_FILE=$O/tune.json
for v in 1 0; do
  OTVM_PPM_ALGEBRA=$v timeout 300 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p_a$v.json 2> $O/bench_1080p_a$v.err
  OTVM_PPM_ALGEBRA=$v timeout 300 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline > $O/bench_480p_a$v.json 2> $O/bench_480p_a$v.err
  python -c "import json;a=json.load(open('$O/bench_480p_a$v.json'));b=json.load(open('$O/bench_1080p_a$v.json'));print('ppm algebra=$v: 480p %.1f fps | 1080p %.2f fps (%.2f ms) conv frac %.3f conv ms %.2f' % (a['value'],b['value'],b['ms_per_step'],b['roofline']['frac'],b['roofline']['conv_ms_per_frame']))"
done
OTVM_PPM_ALGEBRA=1 timeout 300 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p_a1b.json 2>/dev/null
OTVM_PPM_ALGEBRA=0 timeout 300 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p_a0b.json 2>/dev/null
python -c "import json;a=json.load(open('$O/bench_1080p_a1b.json'));b=json.load(open('$O/bench_1080p_a0b.json'));print('repeat: algebra %.2f fps | materialised %.2f fps' % (a['value'],b['value']))"

#!/bin/bash
# round 3: randomised checks over the new routes + every tunable configuration on the real layers; A/B of the fused PPM statistics
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py -m gpu -x -q -k "ppm or sequence_vs_oracle" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python tools/conv_fuzz.py --n 800 --seed 5 2>&1 | tail -3 | tee $O/conv_fuzz.txt
timeout 600 python tools/kernel_fuzz.py --n 150 --seed 5 2>&1 | tail -3 | tee $O/kernel_fuzz.txt
timeout 900 python tools/frame_fuzz.py --n 30 --seed 7 2>&1 | tail -4 | tee $O/frame_fuzz.txt
timeout 900 python tools/tune_verify.py --height 1080 --width 1920 2>&1 | tail -3 | tee $O/tune_verify_1080p.txt
timeout 900 python tools/tune_verify.py --height 480 --width 832 2>&1 | tail -3 | tee $O/tune_verify_480p.txt
timeout 300 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p.json 2> $O/bench_1080p.err
python -c "import json;b=json.load(open('$O/bench_1080p.json'));print('1080p %.2f fps (%.2f ms) conv frac %.3f' % (b['value'],b['ms_per_step'],b['roofline']['frac']))"

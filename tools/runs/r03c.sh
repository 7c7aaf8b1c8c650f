#!/bin/bash
# round 3, batching: new tests first, then the whole suite, then benches at B = 1 / 2 / 4
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py -m gpu -x -q -k "batched or range_guard" -rP > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log; tail -5 $O/pytest_new.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_fullsize.py::test_4k_growing_bank_frame_vs_oracle > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -14 $O/pytest.log
for b in 1 2 4; do
  timeout 600 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --batch $b > $O/bench_480p_b$b.json 2> $O/bench_480p_b$b.err
  python -c "import json;a=json.load(open('$O/bench_480p_b$b.json'));print('480p B=$b: %.1f fps, %.2f ms/step, host %.2f ms, conv frac %.3f' % (a['value'],a['ms_per_step'],a['host_issue_ms_per_frame'],a['roofline']['frac']))" || tail -5 $O/bench_480p_b$b.err
done
for b in 1 2; do
  timeout 600 python bench.py --steps 47 --warmup 3 --no-cpu-baseline --batch $b > $O/bench_1080p_b$b.json 2> $O/bench_1080p_b$b.err
  python -c "import json;a=json.load(open('$O/bench_1080p_b$b.json'));print('1080p B=$b: %.2f fps, %.2f ms/step, host %.2f ms, conv frac %.3f' % (a['value'],a['ms_per_step'],a['host_issue_ms_per_frame'],a['roofline']['frac']))" || tail -5 $O/bench_1080p_b$b.err
done

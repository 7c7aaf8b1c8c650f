#!/bin/bash
# round 4, first lease: the new tests (RCCL with one rank, batched 2-rank tune digests, 480p reference fixture, gn tail stress) + a baseline bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_fullsize.py::test_480p_reference_generated_fixture tests/test_gpu_fullsize.py::test_480p_sequence_vs_oracle "tests/test_gpu_fullsize.py::test_1080p_steady_state_frame_vs_oracle[seed23]" tests/test_gpu_kernels.py::test_gn_table_tail_stress_short -x -q -m gpu -s > gpurun_out/r04a/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r04a/tests.log
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
tail -5 gpurun_out/r04a/tests.log; cat gpurun_out/r04a/bench.json | head -c 600

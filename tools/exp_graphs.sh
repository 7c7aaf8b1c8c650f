cd $GRAFT_REPO_ROOT
export OTVM_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/tune_g.json
for g in 0 1; do
  echo "== GRAPHS=$g"
  OTVM_GRAPHS=$g python tools/host_issue_time.py 2>&1 | grep -v amdgpu
  OTVM_GRAPHS=$g python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-160
  OTVM_GRAPHS=$g python bench.py --steps 47 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-160
done
OTVM_GRAPHS=1 timeout 900 python -m pytest tests/test_gpu_frame.py -x -q -m gpu 2>&1 | tail -5

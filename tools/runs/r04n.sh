#!/bin/bash
# exact row pass of the distance transform with the monotone-minimiser far field + the head kernel walking tiles: tests, isolated
# timings, same-box A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "trimap or edt or distance or head or fuzz_non_conv or reference_vectors or smoke" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -5 $O/tests.log
timeout 300 python tools/glue_bench.py > $O/glue.json 2> $O/glue.err; grep -E "trimap|head" $O/glue.json
timeout 300 python tools/list_gn_passes.py > $O/gn_passes.txt 2>&1; grep -c gn_apply $O/gn_passes.txt
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for rep in 1 2; do
for v in OTVM_HEAD16_WGS=100000 OTVM_HEAD16_WGS=0 OTVM_HEAD16_WGS=768; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'frames/s')" >> $O/ab.txt
done; done
cat $O/ab.txt

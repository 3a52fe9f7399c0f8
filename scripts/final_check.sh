# Round-end sanity pass on one MI355X: every GPU test, smoke(), the default bench line.  usage: bash scripts/final_check.sh <tag>
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $OUT/pytest_gpu.log 2>&1
tail -2 $OUT/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 400 python bench.py > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-600

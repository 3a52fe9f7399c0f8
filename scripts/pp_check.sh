TAG=${1:-pp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_pp.py -m gpu -q --tb=short -x -s -k plugin 2>&1 | tail -40) > $OUT/pytest.log 2>&1; tail -22 $OUT/pytest.log

TAG=${1:-raft}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_raft.py tests/test_gpu_rfc.py tests/test_gpu_pp.py -m gpu -q --tb=short -x -s 2>&1 | tail -40) > $OUT/pytest.log 2>&1; tail -30 $OUT/pytest.log

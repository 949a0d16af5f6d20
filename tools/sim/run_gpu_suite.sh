cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/suite/pytest.log 2>&1
tail -30 gpurun_out/suite/pytest.log
( time python bench.py ) > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err
tail -c 3000 gpurun_out/suite/bench.json; tail -5 gpurun_out/suite/bench.err

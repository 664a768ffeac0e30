cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2ab
timeout -k 10 900 python -m pytest tests -m gpu -q -s > gpurun_out/r2ab/pytest.log 2>&1; echo "pytest rc $?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2ab/pytest.log | head -60

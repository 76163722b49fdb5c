cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_shared.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
echo "--- loop"; SMOE_SHARED_ONE_LAUNCH=0 timeout -k 10 120 python scripts/bench_shared.py --cpu-iters 0 2>/dev/null | cut -c1-300
echo "--- one launch"; timeout -k 10 120 python scripts/bench_shared.py --cpu-iters 0 2>/dev/null | cut -c1-300
echo "--- one launch, clocks"; SMOE_HIP_LIBRARY=$GRAFT_REPO_ROOT/scratch_ab/lib_clk.so timeout -k 10 120 python scripts/bench_shared.py --cpu-iters 0 2>&1 | grep -v "amdgpu.ids" | tail -4 | cut -c1-200

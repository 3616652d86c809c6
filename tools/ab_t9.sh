python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cout1 or cin1 or edge or conv1" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t9 -- python3 $GRAFT_REPO_ROOT/bench.py --workload edsr_x8 --train-only --no-roofline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/prof_summary.py gpurun_out/prof_t9 9 | grep "cout1\|cin1\|total"; rm -rf gpurun_out/prof_t9
bash tools/ab_libs.sh edsr_x8 2 40 - libsrhip_t3.so

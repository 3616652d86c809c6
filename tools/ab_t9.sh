cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t9 -- python3 $GRAFT_REPO_ROOT/tools/mb_tnb3.py --one > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/prof_summary.py gpurun_out/prof_t9 8; rm -rf gpurun_out/prof_t9
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t9 -- python3 $GRAFT_REPO_ROOT/bench.py --workload edsr_x8 --train-only --no-roofline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/prof_summary.py gpurun_out/prof_t9 12; rm -rf gpurun_out/prof_t9

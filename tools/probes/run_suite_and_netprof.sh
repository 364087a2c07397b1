python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r04_gputest.txt; tail -3 gpurun_out/r04_gputest.txt
cd /tmp && export TMPDIR=/tmp
for n in E Dco; do rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/np_$n -- python $GRAFT_REPO_ROOT/tools/net_profile.py $n > $GRAFT_REPO_ROOT/gpurun_out/np_$n.log 2>&1; done
cd $GRAFT_REPO_ROOT; python tools/bench_blur_conv.py 2>&1 | tail -12

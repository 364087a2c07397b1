# round 6: the whole GPU suite + the eager comparator re-measured from the committed find-db
set -x
cd /root/repo
mkdir -p gpurun_out/r06b
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06b/gputest.txt 2>&1
echo "rc=$?" >> gpurun_out/r06b/gputest.txt
tail -15 gpurun_out/r06b/gputest.txt
IDEAS_ROUND=r06 timeout 1500 bash tools/eager_cached.sh 64 400 700 > gpurun_out/r06b/eager.log 2>&1
tail -20 gpurun_out/r06b/eager.log
cp gpurun_out/r06_eager_full.json gpurun_out/r06b/ 2>/dev/null
rm -f gpurun_out/miopen_cache.tar gpurun_out/miopen_db.tar

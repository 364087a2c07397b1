set -x
cd /root/repo
mkdir -p gpurun_out/r06c
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06c/gputest.txt 2>&1
echo "rc=$?" >> gpurun_out/r06c/gputest.txt
tail -15 gpurun_out/r06c/gputest.txt

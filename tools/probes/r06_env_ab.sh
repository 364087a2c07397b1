cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/ab_step.sh "--steps 32 --warmup 8" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"
bash tools/ab_step.sh "--steps 32 --warmup 8 --precision bf16" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"
bash tools/ab_step.sh "--steps 32 --warmup 8 --precision bf16" "IDEAS_SINK_PRIORITY=default" "GPU_MAX_HW_QUEUES=8"

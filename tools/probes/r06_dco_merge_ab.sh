cd /root/repo
timeout 600 python -m pytest tests -m gpu -q -k "forward_pair or test_cooccurrence or step_replay_gpu" 2>&1 | tail -3
bash tools/ab_step.sh "--steps 32 --warmup 8" "IDEAS_DCO_MERGE=0" "IDEAS_DCO_MERGE=2"
bash tools/ab_step.sh "--steps 32 --warmup 8 --precision bf16" "IDEAS_DCO_MERGE=0" "IDEAS_DCO_MERGE=2"

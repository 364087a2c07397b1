# same-box A/B: the driver's launch line with one rank, plain (no process group) against IDEAS_DDP_FORCE_COLLECTIVE=1 (every gradient
# exchange a real RCCL all-reduce over one rank): what the data-parallel plumbing itself costs an iteration
cd ${GRAFT_REPO_ROOT:-/root/repo}
O="--cpu-baseline skip --roofline off --also-bf16 off --steps 16 --warmup 5"
for rep in 1 2; do
  for F in 0 1; do
    echo -n "FORCE_COLLECTIVE=$F (run $rep): "
    IDEAS_DDP_FORCE_COLLECTIVE=$F python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29540 + rep * 2 + F)) bench.py --gpus 1 $O 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['dist_backend'])"
  done
done
for rep in 1 2; do
  for F in 0 1; do
    echo -n "bf16 FORCE_COLLECTIVE=$F (run $rep): "
    IDEAS_DDP_FORCE_COLLECTIVE=$F python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29560 + rep * 2 + F)) bench.py --gpus 1 --precision bf16 $O 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['dist_backend'])"
  done
done

cd ${GRAFT_REPO_ROOT:-/root/repo}
P=29580
for M in none gloo nccl_lazy nccl_used nccl_eager none; do
  P=$((P + 1))
  MODE=$M python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P tools/probes/pg_init_overhead.py 2>&1 | grep "^MODE"
done
# and the same process without the launcher (no OMP_NUM_THREADS=1, no elastic agent beside it)
MODE=none python tools/probes/pg_init_overhead.py 2>&1 | grep "^MODE"
MODE=none OMP_NUM_THREADS=1 python tools/probes/pg_init_overhead.py 2>&1 | grep "^MODE"

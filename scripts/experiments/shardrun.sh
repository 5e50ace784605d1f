export HG_BENCH_BACKEND=gloo HG_BENCH_SAME_DEVICE=1
for v in "HG_OPEN_AT=0" "HG_OPEN_AT=2" "HG_OPEN_AT=2 HG_EQ_ONE_LAUNCH=1" "HG_OPEN_AT=2 HG_NO_GRAPH=1"; do
echo "== $v"
env $v python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --ring-degree 4096 --crt-moduli 2 --no-cpu-baseline 2>&1 | grep -v "^W\|^\*\*\*\|^$" | tail -4 | cut -c1-300
done

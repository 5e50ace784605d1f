export HG_BENCH_BACKEND=gloo HG_BENCH_SAME_DEVICE=1 HG_OPEN_AT=2
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2
A="bench.py --gpus 2 --steps 2 --warmup 1 --ring-degree 4096 --crt-moduli 2 --no-cpu-baseline"
RANK=1 LOCAL_RANK=1 python $A > /tmp/r1.log 2>&1 &
RANK=0 LOCAL_RANK=0 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV stop" -ex run -ex bt --args python $A 2>&1 | grep -v "^\[New\|^\[Thread\|^warning" | tail -40
wait
tail -5 /tmp/r1.log

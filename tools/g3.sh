export TMPDIR=/tmp
python -m pytest tests/test_multi_gloo.py -m gpu -x -q 2>&1 | tail -5
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --workload cfg2 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400

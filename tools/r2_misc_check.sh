#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
# the rank plumbing as the driver uses it: torch.distributed.run with one rank, and the spawn path asking for more GPUs than exist
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu | cut -c1-200 || exit 1
python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu; echo "exit code $? (2 expected: one GPU here)"
python -c "import __graft_entry__ as g; g.smoke()"

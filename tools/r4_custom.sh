#!/bin/bash
# scratch step of tools/r4_visit.sh
OUT=$1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-baseline-seconds 0 --no-other-configs 2>/dev/null | cut -c1-400

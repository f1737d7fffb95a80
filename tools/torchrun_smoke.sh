#!/bin/bash
# launch contract check on one GPU: the exact command shape the driver uses for N>1, with N=1
# (--force-reducer: the GradReducer's bucket bookkeeping runs too; no collectives at world 1)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --force-reducer

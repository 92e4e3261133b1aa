#!/bin/bash
# Rehearsal of the multi-GPU bench modes on a one-GPU box: each mode with one rank, the slab mode once more through the
# collectives of a one-rank RCCL group, and both modes with two ranks over gloo (host-staged) on the same GPU.
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/bench_modes
mkdir -p $OUT
timeout -k 10 200 python -u bench.py --config slab32768 --size 4096 --steps 10 --warmup 2 > $OUT/slab4096_1.json
echo "slab 4096 one rank ok"
timeout -k 10 200 python -u bench.py --config slab32768 --size 4096 --steps 10 --warmup 2 --force-exchange > $OUT/slab4096_1_nccl.json
echo "slab 4096 one rank, forced RCCL exchange in tile groups ok"
timeout -k 10 200 python -u bench.py --config slab32768 --size 4096 --steps 10 --warmup 2 --force-exchange --exchange-groups 1 > $OUT/slab4096_1_nccl_g1.json
echo "slab 4096 one rank, forced RCCL exchange in one piece ok"
timeout -k 10 200 python -u bench.py --config batch1024rgb --planes 96 --batch 32 --steps 10 --warmup 2 > $OUT/batch96_1.json
echo "batch 96 planes one rank ok"
DDN_BENCH_BACKEND=gloo timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --config slab32768 --size 2048 --steps 4 --warmup 1 > $OUT/slab2048_2_gloo.json
echo "slab 2048 two ranks gloo ok"
DDN_BENCH_BACKEND=gloo timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --config batch1024rgb --planes 48 --batch 16 --steps 10 --warmup 2 > $OUT/batch48_2_gloo.json
echo "batch 48 planes two ranks gloo ok"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_modes/slab*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(d["ms_per_step"], 3), d["final_loss"][:2], d["stage_ms"], d["config"]["exchange_tile_groups"])
PY

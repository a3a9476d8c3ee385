#!/bin/bash
# 2-rank bench on ONE GPU (gloo, both ranks on device 0): functional check of tile ownership + reduce + Output;
# the frame hash must equal the 1-rank frame hash.
export GPT_BENCH_BACKEND=gloo GPT_BENCH_SHARE_GPU=1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/one_rank.json 2> gpurun_out/one_rank.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/two_rank.json 2> gpurun_out/two_rank.err
tail -2 gpurun_out/two_rank.err
python - <<'PY'
import json
a = json.loads(open("gpurun_out/one_rank.json").read().strip().splitlines()[-1])
b = json.loads(open("gpurun_out/two_rank.json").read().strip().splitlines()[-1])
print("1 rank :", a["value"], a["config"]["accumulator_sha1"], a["n_gpus"])
print("2 ranks:", b["value"], b["config"]["accumulator_sha1"], b["n_gpus"], "(both on one GPU: no speedup expected)")
print("FRAME IDENTICAL" if a["config"]["accumulator_sha1"] == b["config"]["accumulator_sha1"] else "FRAME DIFFERS")
PY

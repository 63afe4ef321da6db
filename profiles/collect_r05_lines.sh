#!/bin/bash
# The other round-5 bench lines, the hard-scene tables and the parity campaigns on the library collect_r05.sh stamped.
# Run on the GPU box from the repo root; everything lands in gpurun_out/r05l/ (copy what is to be judged into profiles/).
O=gpurun_out/r05l
mkdir -p $O
B="--steps 5 --warmup 2 --no-cpu-baseline --no-extras"
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --config 3 $B > $O/bench_dual.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --resolution low $B > $O/bench_lowres.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --config 2 $B > $O/bench_config2.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --config 4 $B > $O/bench_config4_1gpu.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --config 5 $B > $O/bench_config5_1gpu.json 2>> $O/bench.err || exit 1
XSW_BENCH_BACKEND=gloo XSW_BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --verify-gather --no-extras --no-cpu-baseline 2>> $O/bench.err | grep '^{' > $O/bench_2rank_rehearsal_gloo_one_device.json || exit 1
timeout -k 10 300 python3 profiles/hard_scenes.py --verify > $O/hard_scenes_default.txt 2>> $O/bench.err || exit 1
XSW_LONG_RUN=0 timeout -k 10 300 python3 profiles/hard_scenes.py > $O/hard_scenes_two_kernel_chain.txt 2>> $O/bench.err || exit 1
for f in $O/bench*.json; do python3 -c "
import sys, json
d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms')"; done
cat $O/hard_scenes_default.txt

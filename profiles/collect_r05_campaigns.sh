#!/bin/bash
# Round-5 parity campaigns (whole rasters vs the exhaustive sweep / XSW_ALGO_EXACT: 0 differing pixels expected) and the random
# soak, on the library collect_r05.sh stamped.  Run on the GPU box from the repo root; everything lands in gpurun_out/r05l/.
O=gpurun_out/r05l
mkdir -p $O
# parity campaigns (whole rasters vs the exhaustive sweep / XSW_ALGO_EXACT: 0 differing pixels expected) and the random soak
timeout -k 10 600 python3 tests/campaign_pruned_vs_exhaustive.py > $O/campaign_pruned_vs_exhaustive.txt 2>> $O/bench.err || exit 1
tail -1 $O/campaign_pruned_vs_exhaustive.txt
timeout -k 10 300 python3 tests/campaign_dual_pruned_vs_exact.py > $O/campaign_dual_pruned_vs_exact.txt 2>> $O/bench.err || exit 1
tail -1 $O/campaign_dual_pruned_vs_exact.txt
XSW_RANDOM_CASES=${XSW_RANDOM_CASES:-12000} timeout -k 10 500 python3 -m pytest tests/test_gpu_kernel.py -m gpu -q -k random_configurations > $O/random_soak.txt 2>&1 || exit 1
tail -1 $O/random_soak.txt

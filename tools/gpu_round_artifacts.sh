#!/bin/bash
# Everything the judged artifacts under profiles/ come from, in one GPU call (each step capped):
# kernel-trace stats of bench.py per workload, TCC traffic passes, SQ passes of the decode kernel.
TAG=${1:-r01d}
for wl in const41 zipf255 uniform256; do timeout -k 5 240 bash tools/gpu_profile.sh $TAG $wl > gpurun_out/prof_${TAG}_$wl.txt 2>&1; done
for wl in const41 zipf255; do timeout -k 5 300 bash tools/gpu_traffic.sh $wl > gpurun_out/traffic_$wl.txt 2>&1; done
timeout -k 5 300 bash tools/gpu_pmc.sh zipf255 > gpurun_out/pmc_zipf255.txt 2>&1
for f in gpurun_out/prof_${TAG}_*.txt gpurun_out/traffic_*.txt gpurun_out/pmc_zipf255.txt; do tail -n 2 $f | cut -c1-300; done

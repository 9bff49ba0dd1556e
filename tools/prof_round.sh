#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root): per-kernel stats of the bench, then separate PMC passes.
# Outputs land in gpurun_out/; copy the ones to be judged into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ks -o r -- python3 $R/bench.py --no-latency > $R/gpurun_out/r01_bench_under_rocprof.txt 2>&1
cp $(find /tmp/p_ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_kernel_stats.csv
# the same with one batch in flight: every kernel runs alone, so these are the per-kernel durations DESIGN.md quotes
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k1 -o r -- python3 $R/bench.py --steps 10 --warmup 2 --depth 1 --no-latency > $R/gpurun_out/r01_bench_depth1_under_rocprof.txt 2>&1
cp $(find /tmp/p_k1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_kernel_stats_depth1.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -o r -- python3 $R/bench.py --steps 4 --warmup 1 --depth 1 --no-cpu-baseline --no-latency > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p_f -name "*counter_collection.csv" | head -1) > $R/gpurun_out/r01_pmc_fetch_size.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -o r -- python3 $R/bench.py --steps 4 --warmup 1 --depth 1 --no-cpu-baseline --no-latency > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p_w -name "*counter_collection.csv" | head -1) > $R/gpurun_out/r01_pmc_write_size.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/p_s -o r -- python3 $R/bench.py --steps 4 --warmup 1 --depth 1 --no-cpu-baseline --no-latency > /dev/null 2>&1 || true
f=$(find /tmp/p_s -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f k_bm_c4 > $R/gpurun_out/r01_pmc_sq_k_bm_c4.txt || true
tail -3 $R/gpurun_out/r01_pmc_fetch_size.txt

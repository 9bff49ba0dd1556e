#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root): per-kernel stats of the bench, separate PMC passes (never combined
# with tracing domains other than --kernel-trace).  Outputs land in gpurun_out/prof_r03/; copy the ones to be judged into profiles/.
#   usage: bash tools/prof_round.sh [kitti|hd1080|uhd4k|calib]...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; mkdir -p $O
ks() {   # ks <tag> <bench args...>: kernel stats CSV + the bench line printed in that run
  local tag=$1; shift; rm -rf /tmp/p_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o r -- python3 $R/bench.py --no-latency --no-e2e --no-cpu-baseline --no-secondary --no-verify "$@" > $O/${tag}_bench.json 2> $O/${tag}_bench.err
  cp $(find /tmp/p_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
}
pmc() {  # pmc <tag> <filter> <counters...> -- <bench args...>
  local tag=$1 filt=$2; shift 2; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift; rm -rf /tmp/q_$tag
  timeout -k 10 400 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d /tmp/q_$tag -o r -- python3 $R/bench.py --no-latency --no-e2e --no-cpu-baseline --no-secondary --no-verify --steps 4 --warmup 1 --depth 1 "$@" > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $(find /tmp/q_$tag -name "*counter_collection.csv" | head -1) "$filt" > $O/${tag}.txt
}
for what in "$@"; do
case $what in
kitti)
  ks r03 ; ks r03_depth1 --steps 10 --warmup 2 --depth 1
  pmc r03_pmc_fetch_size "" FETCH_SIZE -- ; pmc r03_pmc_write_size "" WRITE_SIZE --
  pmc r03_pmc_sq1_k_bm_c4 k_bm_c4 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --
  pmc r03_pmc_sq2_k_bm_c4 k_bm_c4 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --
  pmc r03_pmc_sq_all_kernels "" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVES -- ;;
hd1080|uhd4k|euroc|refdefault)
  ks r03_$what --workload $what ; ks r03_${what}_depth1 --workload $what --depth 1 --steps 10 --warmup 2
  pmc r03_${what}_pmc_sq1_k_bm_c4 k_bm_c4 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- --workload $what
  pmc r03_${what}_pmc_fetch_size k_bm_c4 FETCH_SIZE -- --workload $what ; pmc r03_${what}_pmc_write_size k_bm_c4 WRITE_SIZE -- --workload $what ;;
rowsplit)
  rm -rf /tmp/p_rs; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/p_rs -o rs4 -- python3 $R/tools/rowsplit_probe.py 4 20 serial > $O/r03_rowsplit4_serial.txt 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/p_rs -name "*results.db" | head -1) 30 > $O/r03_rowsplit4_kernel_stats.txt
  timeout -k 10 300 python3 $R/bench.py --workload uhd4k --rowsplit --gpus 1 --steps 100 --warmup 10 > $O/r03_rowsplit_bench.json 2> $O/r03_rowsplit_bench.err ;;
calib)
  rm -rf /tmp/q_cal_f /tmp/q_cal_w
  # the probe is built from its source where it runs (never a checked-in binary), and is the program right after `--`
  hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/tools/probe/fetch_calib.hip || exit 1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/q_cal_f -o r -- /tmp/fetch_calib > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/q_cal_w -o r -- /tmp/fetch_calib > /dev/null 2>&1
  { echo "# 1 GiB streamed in and out per kernel (1048576 KB): FETCH_SIZE / WRITE_SIZE in KB by access width per lane"; python3 $R/tools/pmc_summary.py $(find /tmp/q_cal_f -name "*counter_collection.csv" | head -1) k_copy;
    python3 $R/tools/pmc_summary.py $(find /tmp/q_cal_w -name "*counter_collection.csv" | head -1) k_copy; } > $O/r03_pmc_calibration.txt ;;
esac
done
ls $O | tr '\n' ' '

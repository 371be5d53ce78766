#!/bin/bash
# Stall attribution of kmer_grid_kernel (the Occ-rank kernel): one rocprofv3 --pmc pass per counter group over the same
# command, the seed stage of bench.py at 20k reads per launch over the 100k x 10 kb index.  Run on the GPU box:
#   bash tools/pmc_grid.sh            -> gpurun_out/pmc_grid/<group>/...counter_collection.csv
# (no trace options beside --pmc; the program follows `--` directly)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_grid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
    name=$1; shift
    echo "[pmc_grid] pass $name: $*" | tee -a $OUT/progress.log
    timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-include-regex kmer_grid --output-format csv -d $OUT/$name -- \
        python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --stage seeds --reads-per-step 20000 > $OUT/$name.log 2>&1
    tail -1 $OUT/$name.log | cut -c1-200 | tee -a $OUT/progress.log
}
GROUPS_DEFAULT="sq tcp_stall utcl1 utcl1_stall tcp_lat tcc ta tcp_fifo sq2"
for g in ${PMC_GROUPS:-$GROUPS_DEFAULT}; do
case $g in
sq)          run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD ;;
sq2)         run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM ;;
tcp_stall)   run tcp_stall TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_GATE_EN1 ;;
utcl1)       run utcl1 TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS ;;
utcl1_stall) run utcl1_stall TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS ;;
tcp_lat)     run tcp_lat TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_CACHE_ACCESSES TCP_CACHE_MISS ;;
tcp_fifo)    run tcp_fifo TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCP_TA_ADDR_STALL_CYCLES ;;
tcc)         run tcc TCC_REQ TCC_TAG_STALL TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL ;;
ta)          run ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS ;;
*) echo "unknown group $g"; exit 2 ;;
esac
done
echo "[pmc_grid] done" | tee -a $OUT/progress.log

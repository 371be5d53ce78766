#!/bin/bash
# Counters of the FM-extension kernel of the walk-parallel flow (wp_extend_kernel): one rocprofv3 --pmc pass per group over the
# same command, bench.py's --nodp stage at $READS reads per step over the 100k x 10 kb index.  Run on the GPU box:
#   bash tools/pmc_wp.sh            -> gpurun_out/pmc_wp/<group>/...counter_collection.csv
# (no trace options beside --pmc; the program follows `--` directly; at most two TA_* counters per pass)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_wp
READS=${READS:-10000}
KREGEX=${KREGEX:-wp_extend}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
    name=$1; shift
    echo "[pmc_wp] pass $name: $*" | tee -a $OUT/progress.log
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-include-regex $KREGEX --output-format csv -d $OUT/$name -- \
        python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --streams 1 --stage correct-nodp --reads-per-step $READS > $OUT/$name.log 2>&1
    tail -1 $OUT/$name.log | cut -c1-200 | tee -a $OUT/progress.log
}
for g in ${PMC_GROUPS:-sq sq2 sq3}; do
case $g in
sq)   run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD ;;
sq2)  run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_FLAT ;;
sq3)  run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT ;;
tcp)  run tcp TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_CACHE_ACCESSES TCP_CACHE_MISS ;;
*) echo "unknown group $g"; exit 2 ;;
esac
done
echo "[pmc_wp] done" | tee -a $OUT/progress.log
